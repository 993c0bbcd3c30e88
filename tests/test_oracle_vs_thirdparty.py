"""Pin the CPU oracle against the third-party arithmetic SURVEY.md section 2.2 / 8c cites.

The reference tree has no tests or fixtures (parity unpinned); these checks make sure the
oracle's fp32 restatement agrees with torch.stft and with HF `transformers` T5 modules built
from a LOCAL config (no hub access) carrying the same seeded weights.
"""
import math

import numpy as np
import pytest
import torch

from oracle import ymt3_oracle as O
from yourmt3_amd.config import YMT3Config
from yourmt3_amd.weights import make_weights

CFG = YMT3Config(segment_samples=8191, max_decode_len=32)   # 64 frames: keeps the CPU suite fast


@pytest.fixture(scope="module")
def weights():
    return make_weights(CFG, seed=1234)


def test_power_spectrogram_matches_torch_stft():
    cfg = YMT3Config()
    a = O.synthetic_audio(2, cfg)
    st = torch.stft(a, cfg.n_fft, cfg.hop, window=torch.hann_window(cfg.n_fft), center=True,
                    pad_mode="reflect", return_complex=True)
    ref = (st.real ** 2 + st.imag ** 2).transpose(1, 2)
    got = O.power_spectrogram(a, cfg)
    assert got.shape == (2, 256, 1025)
    assert (ref - got).abs().max() <= 2e-6 * ref.abs().max()


def test_stft_known_answers():
    cfg = YMT3Config()
    # impulse at the centre of frame 8 -> flat spectrum of height window[n_fft/2]^2 = 1
    x = torch.zeros(1, cfg.segment_samples)
    x[0, 8 * cfg.hop] = 1.0
    p = O.power_spectrogram(x, cfg)
    assert torch.allclose(p[0, 8], torch.ones(cfg.n_freqs), atol=1e-5)
    # bin-centred tone: energy (A * N/4)^2 at bin k for a Hann window
    k = 100
    n = torch.arange(cfg.segment_samples, dtype=torch.float64)
    tone = torch.cos(2 * math.pi * k * n / cfg.n_fft).float()[None]
    p = O.power_spectrogram(tone, cfg)
    assert abs(p[0, 100, k].item() - (cfg.n_fft / 4) ** 2) / (cfg.n_fft / 4) ** 2 < 1e-4
    assert p[0, 100].argmax().item() == k


def test_mel_filterbank_shape_and_support():
    fb = O.mel_filterbank_htk(128, 2048, 16000, 50.0, 8000.0)
    assert fb.shape == (128, 1025)
    assert (fb >= 0).all() and fb.max() <= 1.0
    assert ((fb > 0).sum(0) <= 2).all()          # every bin feeds at most two triangles
    assert ((fb > 0).sum(1) > 0).all()           # no empty filter at n_fft 2048
    # interior triangles partition unity between first and last centre
    s = fb.sum(0)
    lo = int(np.ceil(700 * (10 ** ((2595 * np.log10(1 + 50 / 700) + (2595 * np.log10(1 + 8000 / 700) - 2595 * np.log10(1 + 50 / 700)) / 129) / 2595) - 1) / 7.8125))
    assert torch.allclose(s[lo + 1:1000], torch.ones(1000 - lo - 1), atol=1e-5)


def test_relative_position_bucket_matches_hf_exactly():
    from transformers.models.t5.modeling_t5 import T5Attention
    q = torch.arange(512)[:, None]
    k = torch.arange(512)[None, :]
    rel = k - q
    for bidir in (True, False):
        ref = T5Attention._relative_position_bucket(rel, bidirectional=bidir, num_buckets=32, max_distance=128)
        got = O.relative_position_bucket(rel.numpy(), bidir, 32, 128)
        assert np.array_equal(ref.numpy(), got)
    # decoder distances up to the full 1024-token window
    d = -torch.arange(1024)
    ref = T5Attention._relative_position_bucket(d, bidirectional=False, num_buckets=32, max_distance=128)
    assert np.array_equal(ref.numpy(), O.relative_position_bucket(d.numpy(), False, 32, 128))


def _hf_model(W, cfg):
    from transformers import T5Config, T5ForConditionalGeneration
    hc = T5Config(vocab_size=cfg.vocab, d_model=cfg.d_model, d_kv=cfg.d_kv, d_ff=cfg.d_ff,
                  num_layers=cfg.n_enc_layers, num_decoder_layers=cfg.n_dec_layers, num_heads=cfg.n_heads,
                  relative_attention_num_buckets=cfg.rel_buckets, relative_attention_max_distance=cfg.rel_max_distance,
                  feed_forward_proj="relu", dropout_rate=0.0, layer_norm_epsilon=cfg.ln_eps,
                  decoder_start_token_id=cfg.pad_id, pad_token_id=cfg.pad_id, eos_token_id=cfg.eos_id,
                  tie_word_embeddings=False, attn_implementation="eager")
    m = T5ForConditionalGeneration(hc).eval()
    sd = {}
    inner = cfg.inner
    sd["shared.weight"] = W["dec.embed"]
    sd["encoder.embed_tokens.weight"] = W["dec.embed"]
    sd["decoder.embed_tokens.weight"] = W["dec.embed"]
    sd["lm_head.weight"] = W["dec.lm_head"]
    sd["encoder.final_layer_norm.weight"] = W["enc.ln_f"]
    sd["decoder.final_layer_norm.weight"] = W["dec.ln_f"]
    sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"] = W["enc.relbias"]
    sd["decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"] = W["dec.relbias"]
    for l in range(cfg.n_enc_layers):
        p, h = f"enc.{l}.", f"encoder.block.{l}.layer."
        q, k, v = W[p + "wqkv"].split(inner, 0)
        sd[h + "0.SelfAttention.q.weight"], sd[h + "0.SelfAttention.k.weight"], sd[h + "0.SelfAttention.v.weight"] = q, k, v
        sd[h + "0.SelfAttention.o.weight"] = W[p + "wo"]
        sd[h + "0.layer_norm.weight"] = W[p + "ln1"]
        sd[h + "1.DenseReluDense.wi.weight"] = W[p + "wi"]
        sd[h + "1.DenseReluDense.wo.weight"] = W[p + "wo2"]
        sd[h + "1.layer_norm.weight"] = W[p + "ln2"]
    for l in range(cfg.n_dec_layers):
        p, h = f"dec.{l}.", f"decoder.block.{l}.layer."
        q, k, v = W[p + "wqkv"].split(inner, 0)
        sd[h + "0.SelfAttention.q.weight"], sd[h + "0.SelfAttention.k.weight"], sd[h + "0.SelfAttention.v.weight"] = q, k, v
        sd[h + "0.SelfAttention.o.weight"] = W[p + "wo"]
        sd[h + "0.layer_norm.weight"] = W[p + "ln1"]
        kc, vc = W[p + "wkv_c"].split(inner, 0)
        sd[h + "1.EncDecAttention.q.weight"] = W[p + "wq_c"]
        sd[h + "1.EncDecAttention.k.weight"], sd[h + "1.EncDecAttention.v.weight"] = kc, vc
        sd[h + "1.EncDecAttention.o.weight"] = W[p + "wo_c"]
        sd[h + "1.layer_norm.weight"] = W[p + "ln2"]
        sd[h + "2.DenseReluDense.wi.weight"] = W[p + "wi"]
        sd[h + "2.DenseReluDense.wo.weight"] = W[p + "wo2"]
        sd[h + "2.layer_norm.weight"] = W[p + "ln3"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("embed_tokens" in k or "shared" in k for k in missing), missing
    # transformers 5.x always ties lm_head to `shared` (TP: configuration_t5.py:82-83); this build's
    # head is untied, so give lm_head its own Parameter after loading and re-load the embedding.
    assert hc.scale_decoder_outputs is False
    m.lm_head.weight = torch.nn.Parameter(W["dec.lm_head"].clone())
    with torch.no_grad():
        m.shared.weight.copy_(W["dec.embed"])
    assert torch.equal(m.decoder.embed_tokens.weight, W["dec.embed"])
    assert torch.equal(m.lm_head.weight, W["dec.lm_head"])
    return m


@pytest.fixture(scope="module")
def hf(weights):
    return _hf_model(weights, CFG)


def test_rmsnorm_matches_t5layernorm(weights):
    from transformers.models.t5.modeling_t5 import T5LayerNorm
    ln = T5LayerNorm(CFG.d_model, eps=CFG.ln_eps)
    ln.weight.data.copy_(weights["enc.0.ln1"])
    x = torch.randn(3, 7, CFG.d_model, generator=torch.Generator().manual_seed(5)) * 3
    assert torch.allclose(ln(x), O.rmsnorm(x, weights["enc.0.ln1"], CFG.ln_eps), atol=1e-6, rtol=1e-6)


def test_encoder_matches_hf_t5stack_fp32(weights, hf):
    a = O.synthetic_audio(2, CFG)
    mel = O.logmel(a, CFG)
    h0 = O.input_projection(mel, weights, bf16=False)
    got = O.encoder_t5(h0, weights, CFG, bf16=False)
    with torch.no_grad():
        ref = hf.encoder(inputs_embeds=h0).last_hidden_state
    assert got.shape == ref.shape == (2, CFG.n_frames, CFG.d_model)
    assert (got - ref).abs().max().item() < 2e-4


def test_greedy_decode_matches_hf_generate_fp32(weights, hf):
    a = O.synthetic_audio(2, CFG)
    _, enc = O.encode(a, weights, CFG, bf16=False)
    n = 24
    toks, logits = O.greedy_decode(enc, weights, CFG, n, bf16=False, return_logits=True)
    from transformers.modeling_outputs import BaseModelOutput
    with torch.no_grad():
        out = hf.generate(encoder_outputs=BaseModelOutput(last_hidden_state=enc), max_new_tokens=n, min_new_tokens=n,
                          do_sample=False, num_beams=1, output_logits=True, return_dict_in_generate=True)
    ref_tokens = out.sequences[:, 1:]                     # drop decoder_start
    ref_logits = torch.stack(out.logits, 1)
    assert (logits[:, 0] - ref_logits).abs().max().item() < 5e-4
    assert torch.equal(toks[:, 0].long(), ref_tokens)
    assert len(set(toks[0, 0].tolist())) > 8              # the stream is not a collapsed fixed point


def test_decoder_matches_hf_at_every_position_of_a_1024_token_decode(weights):
    """The oracle's cached step-by-step decode (self-attention cache append, unidirectional bias by distance with the past
    offset: TP modeling_t5.py:264-279, 281-369) against HF's one-shot full-sequence decoder forward (causal mask, bias table for
    all (query, key) pairs) at EVERY position of a 1024-token stream -- far past the bucket saturation (distance >= 128) and
    the positions the HIP self-attention's long-sequence paths start at (383).  fp32 both sides: agreement to round-off."""
    from transformers.modeling_outputs import BaseModelOutput
    cfg = CFG.with_(max_decode_len=1024, eos_id=-1)
    hf_long = _hf_model(weights, cfg.with_(eos_id=1))          # (eos only configures HF's generate; forward ignores it)
    _, enc = O.encode(O.synthetic_audio(1, cfg), weights, cfg, bf16=False)
    toks, logits = O.greedy_decode(enc, weights, cfg, 1024, bf16=False, return_logits=True)
    dec_in = torch.cat([torch.full((1, 1), cfg.pad_id, dtype=torch.long), toks[:, 0, :-1].long()], 1)
    with torch.no_grad():
        ref = hf_long(encoder_outputs=BaseModelOutput(last_hidden_state=enc), decoder_input_ids=dec_in).logits     # (1, 1024, V)
    err = (ref - logits[:, 0]).abs().amax(-1)[0]
    assert err.max().item() < 1e-4, (err.max().item(), int(err.argmax()))
    assert torch.equal(ref.argmax(-1)[0], toks[0, 0].long())
    assert len(set(toks[0, 0, 512:].tolist())) > 30          # still a varied stream late in the decode


def test_eos_fill_matches_hf(weights, hf):
    # force an early EOS by making the EOS row of the head dominant from step 3 on is hard with random
    # weights; instead check the fill rule directly on a crafted logits sequence via the oracle loop.
    cfg = CFG.with_(eos_id=1)
    a = O.synthetic_audio(1, cfg)
    _, enc = O.encode(a, weights, cfg, bf16=False)
    W2 = dict(weights)
    toks_free = O.greedy_decode(enc, W2, cfg.with_(eos_id=-1), 12, bf16=False)
    eos_tok = int(toks_free[0, 0, 4])
    cfg2 = cfg.with_(eos_id=eos_tok)                      # declare the 5th emitted id to be EOS
    toks = O.greedy_decode(enc, W2, cfg2, 12, bf16=False)[0, 0].tolist()
    first = toks.index(eos_tok)
    assert first <= 4
    assert all(t == cfg2.pad_id for t in toks[first + 1:])
    assert toks[:first + 1] == toks_free[0, 0, :first + 1].tolist()
