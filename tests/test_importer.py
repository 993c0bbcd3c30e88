"""Importer: HF-T5-named weights -> this build's tensors.  CPU: the imported weights reproduce
HF's own outputs through the oracle.  GPU: the HIP path on imported weights against HF T5 itself (fp32), i.e. the
third-party arithmetic directly, not only this build's oracle."""
import pytest
import torch

from oracle import ymt3_oracle as O
from yourmt3_amd.config import YMT3Config
from yourmt3_amd.importer import from_t5_state_dict

CFG = YMT3Config(segment_samples=8191, max_decode_len=32)


def _hf(seed=3):
    from transformers import T5Config, T5ForConditionalGeneration
    torch.manual_seed(seed)
    hc = T5Config(vocab_size=CFG.vocab, d_model=CFG.d_model, d_kv=CFG.d_kv, d_ff=CFG.d_ff, num_layers=CFG.n_enc_layers,
                  num_decoder_layers=CFG.n_dec_layers, num_heads=CFG.n_heads, feed_forward_proj="relu", dropout_rate=0.0,
                  layer_norm_epsilon=CFG.ln_eps, decoder_start_token_id=CFG.pad_id, pad_token_id=CFG.pad_id, eos_token_id=CFG.eos_id,
                  tie_word_embeddings=False, attn_implementation="eager")
    m = T5ForConditionalGeneration(hc).eval()
    with torch.no_grad():                     # untie + give the head and embeddings some spread, then make everything bf16-exact
        m.lm_head.weight = torch.nn.Parameter(torch.randn(CFG.vocab, CFG.d_model) * CFG.d_model ** -0.5)
        m.shared.weight.copy_(torch.randn(CFG.vocab, CFG.d_model) * 4.0)
        for p in m.parameters():
            if p.dim() == 2 and p.shape != (CFG.rel_buckets, CFG.n_heads):
                p.copy_(p.to(torch.bfloat16).float())
    return m


def _imported(m):
    sd = dict(m.state_dict())
    sd["lm_head.weight"] = m.lm_head.weight
    g = torch.Generator().manual_seed(1)
    return from_t5_state_dict(sd, CFG, in_proj_w=torch.randn(CFG.d_model, CFG.n_mels, generator=g) * 0.02,
                              in_proj_b=torch.zeros(CFG.d_model))


def test_imported_weights_reproduce_hf_through_the_oracle():
    m = _hf()
    W = _imported(m)
    a = O.synthetic_audio(2, CFG)
    h0 = O.input_projection(O.logmel(a, CFG), W, bf16=False)
    enc = O.encoder_t5(h0, W, CFG, bf16=False)
    with torch.no_grad():
        ref = m.encoder(inputs_embeds=h0).last_hidden_state
    assert (enc - ref).abs().max().item() < 2e-4
    from transformers.modeling_outputs import BaseModelOutput
    with torch.no_grad():
        out = m.generate(encoder_outputs=BaseModelOutput(last_hidden_state=enc), max_new_tokens=12, min_new_tokens=12,
                         do_sample=False, num_beams=1)
    assert torch.equal(O.greedy_decode(enc, W, CFG, 12, bf16=False)[:, 0].long(), out[:, 1:])


def test_importer_rejects_mismatched_shapes():
    m = _hf()
    sd = dict(m.state_dict())
    sd["lm_head.weight"] = m.lm_head.weight
    with pytest.raises(ValueError):
        from_t5_state_dict(sd, CFG.with_(vocab=1024))
    sd["encoder.final_layer_norm.weight"] = torch.full((CFG.d_model,), float("nan"))
    with pytest.raises(ValueError):
        from_t5_state_dict(sd, CFG)


@pytest.mark.gpu
def test_hip_path_on_imported_weights_matches_hf_t5_directly():
    from yourmt3_amd.model import YourMT3
    m = _hf()
    W = _imported(m)
    model = YourMT3(CFG, W, device=0, max_batch=2)
    a = O.synthetic_audio(2, CFG)
    mel = model.logmel(a.cuda())
    enc = model.encode(mel)
    h0 = O.input_projection(mel.cpu(), W, bf16=False)
    with torch.no_grad():
        ref_enc = m.encoder(inputs_embeds=h0).last_hidden_state
    # bf16 operands vs HF fp32: a few percent of the unit-RMS output
    assert (enc.float().cpu() - ref_enc).abs().max().item() < 0.12
    assert (enc.float().cpu() - ref_enc).abs().mean().item() < 0.01
    n = 10
    from transformers.modeling_outputs import BaseModelOutput
    with torch.no_grad():
        out = m.generate(encoder_outputs=BaseModelOutput(last_hidden_state=enc.float().cpu()), max_new_tokens=n, min_new_tokens=n,
                         do_sample=False, num_beams=1, output_logits=True, return_dict_in_generate=True)
    hf_tokens = out.sequences[:, 1:]
    hf_logits = torch.stack(out.logits, 1)
    got_t, got_l = model.decode(enc, n, forced=hf_tokens[:, None, :].int().cuda(), return_logits=True)
    d = (got_l.cpu()[:, 0] - hf_logits).abs()
    assert d.max().item() < 0.25 and d.mean().item() < 0.03
    top2 = hf_logits.topk(2, -1).values
    safe = (top2[..., 0] - top2[..., 1]) > 0.5
    assert safe.any() and torch.equal(got_t.cpu()[:, 0][safe], hf_tokens[safe].int())
    model.close()


def test_lightning_shaped_checkpoint_container(tmp_path):
    """the `.ckpt` container (state_dict under a key, module-path prefixes, non-tensor metadata), read with a
    loader that executes nothing from the file; dims are checked against the config; a file the safe loader refuses is an error.
    (No real YourMT3 checkpoint exists offline: the file is written here, in that shape, from a random HF T5.)"""
    from yourmt3_amd.importer import from_checkpoint, load_checkpoint_tensors, find_t5_prefix
    m = _hf()
    direct = _imported(m)
    sd = {"model." + k: v.clone() for k, v in m.state_dict().items()}
    sd["model.lm_head.weight"] = m.lm_head.weight.detach().clone()
    g = torch.Generator().manual_seed(1)
    sd["model.pre_encoder.proj.weight"] = torch.randn(CFG.d_model, CFG.n_mels, generator=g) * 0.02
    sd["model.pre_encoder.proj.bias"] = torch.zeros(CFG.d_model)
    ckpt = {"state_dict": sd, "epoch": 7, "global_step": 12345, "pytorch-lightning_version": "2.1.0",
            "hyper_parameters": {"lr": 1e-3, "task": "mt3_full"}, "optimizer_states": []}
    path = str(tmp_path / "model.ckpt")
    torch.save(ckpt, path)
    assert find_t5_prefix(load_checkpoint_tensors(path)) == "model."
    W = from_checkpoint(path, CFG, in_proj_names=("model.pre_encoder.proj.weight", "model.pre_encoder.proj.bias"))
    assert W.keys() == direct.keys() and all(torch.equal(W[k], direct[k]) for k in W)
    bare = str(tmp_path / "bare.pt")
    torch.save({k[len("model."):]: v for k, v in sd.items()}, bare)             # a bare state dict, no prefix
    with pytest.raises(ValueError, match="in_proj_names"):                        # no silent all-zero encoder input
        from_checkpoint(bare, CFG)
    W2 = from_checkpoint(bare, CFG, allow_zero_in_proj=True)
    assert torch.equal(W2["dec.3.wkv_c"], direct["dec.3.wkv_c"]) and float(W2["in_proj.w"].abs().max()) == 0.0
    W3 = from_checkpoint(bare, CFG, name_map={"in_proj.w": "pre_encoder.proj.weight", "in_proj.b": "pre_encoder.proj.bias"})   # the same through the table
    assert torch.equal(W3["in_proj.w"], direct["in_proj.w"])
    with pytest.raises(ValueError, match="d_ff"):
        from_checkpoint(path, CFG.with_(d_ff=1024))
    with pytest.raises(ValueError, match="no T5 encoder"):
        torch.save({"state_dict": {"foo.weight": torch.zeros(2)}}, str(tmp_path / "x.ckpt"))
        from_checkpoint(str(tmp_path / "x.ckpt"), CFG)

    class Evil:                                                                    # a pickled class: the safe loader must refuse it
        def __reduce__(self):
            return (print, ("executed from the checkpoint",))
    torch.save({"state_dict": sd, "callbacks": Evil()}, str(tmp_path / "evil.ckpt"))
    with pytest.raises(ValueError, match="safe loader"):
        from_checkpoint(str(tmp_path / "evil.ckpt"), CFG)


def test_ymt3_plus_shaped_container_round_trips_through_the_rule_table(tmp_path):
    """The importer is a table (importer.t5_rules + extra_rules): HF T5 names for the T5 tensors, this build's names under `ymt3.` for what
    T5 has no name for -- the Perceiver-TF encoder, the channel embedding, the MoE router and experts.  A YMT3+-shaped set of weights
    (Perceiver-TF encoder + 13 channels + MoE decoder) written as a Lightning-shaped container comes back tensor for tensor; the same
    container imports into the fp8 config (experts quantised at import); a missing tensor is named."""
    from yourmt3_amd.config import ENC_PERCEIVER_TF, FFN_MOE
    from yourmt3_amd.importer import extra_rules, from_checkpoint, t5_rules, to_checkpoint
    from yourmt3_amd.weights import make_weights
    cfg = YMT3Config(segment_samples=8191, max_decode_len=32, encoder_type=ENC_PERCEIVER_TF, n_enc_layers=0, n_latents=32, ptf_blocks=1,
                     n_channels=13, dec_ffn=FFN_MOE)
    W = make_weights(cfg)
    assert {r[0] for r in t5_rules(cfg) + extra_rules(cfg)} == set(W)                     # every tensor of the blob has a rule
    path = str(tmp_path / "plus.ckpt")
    to_checkpoint(W, cfg, path)
    back = from_checkpoint(path, cfg)
    assert back.keys() == W.keys() and all(torch.equal(back[k], W[k]) for k in W)
    cfg8 = cfg.with_(moe_fp8=1)
    W8, back8 = make_weights(cfg8), from_checkpoint(path, cfg8)
    assert back8.keys() == W8.keys() and all(torch.equal(back8[k], W8[k]) for k in W8)
    st = str(tmp_path / "plus.safetensors")
    to_checkpoint(W, cfg, st)
    assert all(torch.equal(v, W[k]) for k, v in from_checkpoint(st, cfg).items())
    sd = torch.load(path, weights_only=True)["state_dict"]
    del sd["model.ymt3.dec.2.router"]
    torch.save({"state_dict": sd}, str(tmp_path / "short.ckpt"))
    with pytest.raises(ValueError, match="dec.2.router"):
        from_checkpoint(str(tmp_path / "short.ckpt"), cfg)
