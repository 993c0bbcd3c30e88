"""CPU oracle: a plain fp32 PyTorch/numpy restatement of the transcription hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under yourmt3_amd/ imports this module; only tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg use it, and only as the checker /
reported baseline, never as the product path.

PARITY UNPINNED with respect to the reference: /root/reference holds README.md + LICENSE and no
code, tests or fixtures.  What this oracle IS pinned against, in
tests/test_oracle_vs_thirdparty.py, is the third-party arithmetic SURVEY section 2.2 cites:

  * STFT                : torch.stft                      (TP: torch/functional.py:508-690)
  * RMS norm            : T5LayerNorm                     (TP: transformers/models/t5/modeling_t5.py:50-72)
  * attention, no 1/sqrt(d) scale, bias added pre-softmax (TP: modeling_t5.py:144-173, 196-197)
  * relative-position bucket / bias                       (TP: modeling_t5.py:216-279)
  * encoder/decoder block and stack                       (TP: modeling_t5.py:435-509, 640-750)
  * cross-attention K/V computed once, self K/V appended   (TP: modeling_t5.py:281-369, cache_utils.py:127-146)
  * lm_head + greedy argmax + EOS->PAD fill               (TP: modeling_t5.py:1042-1047, generation/utils.py:2894-2937)

Two precisions are restated:
  emulate_bf16=False : everything fp32 -- this is what is compared with the HF T5 modules.
  emulate_bf16=True  : the rounding points of the HIP path (DESIGN.md "Numerics contract"):
                       GEMM inputs rounded to bf16, fp32 accumulation, fp32 residual stream.
                       This is what the GPU parity tests compare with.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------------
# front-end: a1 (frame + window + STFT) and a2 (power + mel + log)
# --------------------------------------------------------------------------------------------
def hann_window(n_fft: int) -> Tensor:
    """Periodic Hann, the torch.stft convention (torch.hann_window default)."""
    n = torch.arange(n_fft, dtype=torch.float64)
    return (0.5 - 0.5 * torch.cos(2.0 * math.pi * n / n_fft)).to(torch.float32)


def mel_filterbank_htk(n_mels: int, n_fft: int, sample_rate: int, f_min: float, f_max: float) -> Tensor:
    """(n_mels, n_freqs) triangular HTK-mel filterbank, no area normalisation.

    No implementation exists in the container (SURVEY section 2.2 star-2: torchaudio/librosa absent), so this
    states the published HTK formula m = 2595 log10(1 + f/700) directly.  Computed in float64.
    """
    n_freqs = n_fft // 2 + 1
    freqs = np.linspace(0.0, sample_rate / 2.0, n_freqs)
    m_lo = 2595.0 * np.log10(1.0 + f_min / 700.0)
    m_hi = 2595.0 * np.log10(1.0 + f_max / 700.0)
    m_pts = np.linspace(m_lo, m_hi, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - freqs[:, None]              # (n_freqs, n_mels + 2)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))            # (n_freqs, n_mels)
    return torch.from_numpy(fb.T.astype(np.float32).copy())


def frame_audio(audio: Tensor, n_fft: int, hop: int) -> Tensor:
    """(B, S) -> (B, T, n_fft): reflect-pad n_fft/2 each side (torch.stft center=True), hop-strided frames."""
    pad = n_fft // 2
    x = torch.nn.functional.pad(audio[:, None, :], (pad, pad), mode="reflect")[:, 0, :]
    return x.unfold(-1, n_fft, hop)


def power_spectrogram(audio: Tensor, cfg) -> Tensor:
    """(B, S) fp32 -> (B, T, n_freqs) |STFT|^2."""
    frames = frame_audio(audio.float(), cfg.n_fft, cfg.hop) * hann_window(cfg.n_fft)
    spec = torch.fft.rfft(frames, dim=-1)
    return spec.real ** 2 + spec.imag ** 2


def logmel(audio: Tensor, cfg) -> Tensor:
    """(B, S) fp32 audio -> (B, T, n_mels) fp32 natural-log mel power."""
    power = power_spectrogram(audio, cfg)
    fb = mel_filterbank_htk(cfg.n_mels, cfg.n_fft, cfg.sample_rate, cfg.f_min, cfg.f_max)
    mel = power @ fb.T
    return torch.log(torch.clamp(mel, min=cfg.log_floor))


# --------------------------------------------------------------------------------------------
# a5: relative-position buckets (integer work -- must be bit exact)
# --------------------------------------------------------------------------------------------
def relative_position_bucket(rel: np.ndarray, bidirectional: bool, num_buckets: int, max_distance: int) -> np.ndarray:
    """Integer restatement of TP: modeling_t5.py:216-262.  `rel` = key_pos - query_pos (int64 array).

    The large-distance branch uses fp32 log exactly as the cited code does (torch.log on a float32
    tensor, divided by a Python float, multiplied, truncated toward zero).
    """
    rel = np.asarray(rel, dtype=np.int64)
    buckets = np.zeros_like(rel)
    nb = num_buckets
    if bidirectional:
        nb //= 2
        buckets = buckets + (rel > 0).astype(np.int64) * nb
        rp = np.abs(rel)
    else:
        rp = -np.minimum(rel, 0)
    max_exact = nb // 2
    is_small = rp < max_exact
    with np.errstate(divide="ignore"):
        lg = np.log(rp.astype(np.float32) / np.float32(max_exact)).astype(np.float32)
    scaled = (lg / np.float32(math.log(max_distance / max_exact))).astype(np.float32) * np.float32(nb - max_exact)
    large = max_exact + np.where(np.isfinite(scaled), scaled, 0).astype(np.int64)
    large = np.minimum(large, nb - 1)
    return buckets + np.where(is_small, rp, large)


def encoder_bias_by_offset(relbias: Tensor, T: int, cfg) -> Tensor:
    """(H, 2T-1) fp32: bias for key_pos - query_pos = off - (T-1), off in [0, 2T-1)."""
    off = np.arange(-(T - 1), T)
    b = relative_position_bucket(off, True, cfg.rel_buckets, cfg.rel_max_distance)
    return relbias[torch.from_numpy(b)].T.contiguous()


def decoder_bias_by_distance(relbias: Tensor, L: int, cfg) -> Tensor:
    """(H, L) fp32: bias for a key `dist` = query_pos - key_pos >= 0 positions back (causal)."""
    b = relative_position_bucket(-np.arange(L), False, cfg.rel_buckets, cfg.rel_max_distance)
    return relbias[torch.from_numpy(b)].T.contiguous()


# --------------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------------
def _r(x: Tensor, on: bool) -> Tensor:
    """bf16 round-to-nearest-even rounding point (identity in the fp32 restatement).  The result keeps x's dtype: run on float64
    tensors (weights and inputs cast to double) the whole oracle accumulates in double with the SAME rounding points -- a second
    evaluation order, used by tests to measure how far two correct bf16 implementations of one model legitimately differ."""
    return x.to(torch.bfloat16).to(x.dtype if x.dtype == torch.float64 else torch.float32) if on else x


def rmsnorm(x: Tensor, gain: Tensor, eps: float) -> Tensor:
    """TP: modeling_t5.py:50-72 -- no mean subtraction, fp32 accumulation."""
    var = (x if x.dtype == torch.float64 else x.float()).pow(2).mean(-1, keepdim=True)
    return x * torch.rsqrt(var + eps) * gain


def split_heads(x: Tensor, H: int) -> Tensor:
    B, T, _ = x.shape
    return x.view(B, T, H, -1).transpose(1, 2)           # (B, H, T, d_kv)


def merge_heads(x: Tensor) -> Tensor:
    B, H, T, D = x.shape
    return x.transpose(1, 2).reshape(B, T, H * D)


def attention(q: Tensor, k: Tensor, v: Tensor, bias: Optional[Tensor], bf16: bool, round_p: bool) -> Tensor:
    """softmax(q k^T + bias) v with NO 1/sqrt(d) scale (TP: modeling_t5.py:196-197).

    round_p=True restates the MFMA encoder kernel: e = exp(s - max) is rounded to bf16 before the
    P.V product while the normaliser sums the unrounded e.  round_p=False is the decode kernels'
    all-fp32 softmax.  Output rounded to bf16 when emulating.
    """
    s = q @ k.transpose(-1, -2)
    if bias is not None:
        s = s + bias
    m = s.max(-1, keepdim=True).values
    e = torch.exp(s - m)
    l = e.sum(-1, keepdim=True)
    if bf16 and round_p:
        o = (_r(e, True) @ v) / l
    else:
        o = (e @ v) / l
    return _r(o, bf16)


# --------------------------------------------------------------------------------------------
# a3/a4: input projection + T5 encoder
# --------------------------------------------------------------------------------------------
def input_projection(mel: Tensor, W: Dict[str, Tensor], bf16: bool) -> Tensor:
    return _r(mel, bf16) @ W["in_proj.w"].T + W["in_proj.b"]


def encoder_t5(h: Tensor, W: Dict[str, Tensor], cfg, bf16: bool) -> Tensor:
    """(B, T, d) fp32 residual stream -> (B, T, d) encoder output (after final norm).

    Block order TP: modeling_t5.py:435-509; bias table shared by all layers (:739-742); final norm (:744).
    """
    B, T, d = h.shape
    H, inner = cfg.n_heads, cfg.inner
    bias_off = encoder_bias_by_offset(W["enc.relbias"], T, cfg)              # (H, 2T-1)
    idx = (torch.arange(T)[None, :] - torch.arange(T)[:, None]) + (T - 1)    # key - query + T-1
    bias = bias_off[:, idx][None]                                            # (1, H, T, T)
    for l in range(cfg.n_enc_layers):
        p = f"enc.{l}."
        xn = _r(rmsnorm(h, W[p + "ln1"], cfg.ln_eps), bf16)
        qkv = _r(xn @ W[p + "wqkv"].T, bf16)
        q, k, v = (split_heads(t, H) for t in qkv.split(inner, dim=-1))
        a = merge_heads(attention(q, k, v, bias, bf16, round_p=True))
        h = h + a @ W[p + "wo"].T
        xn = _r(rmsnorm(h, W[p + "ln2"], cfg.ln_eps), bf16)
        ff = _r(torch.relu(xn @ W[p + "wi"].T), bf16)
        h = h + ff @ W[p + "wo2"].T
    return _r(rmsnorm(h, W["enc.ln_f"], cfg.ln_eps), bf16)


def encode(audio: Tensor, W: Dict[str, Tensor], cfg, bf16: bool) -> Tuple[Tensor, Tensor]:
    mel = logmel(audio, cfg)
    if getattr(cfg, "encoder_type", 0) == 1:
        from oracle.perceiver_oracle import encoder_perceiver_tf
        return mel, encoder_perceiver_tf(mel, W, cfg, bf16)
    return mel, encoder_t5(input_projection(mel, W, bf16), W, cfg, bf16)


# --------------------------------------------------------------------------------------------
# a6: cross-attention K/V, computed once per segment (TP: modeling_t5.py:319-332)
# --------------------------------------------------------------------------------------------
def cross_kv(enc_out: Tensor, W: Dict[str, Tensor], cfg, bf16: bool) -> List[Tuple[Tensor, Tensor]]:
    out = []
    for l in range(cfg.n_dec_layers):
        kv = _r(enc_out @ W[f"dec.{l}.wkv_c"].T, bf16)
        k, v = kv.split(cfg.inner, dim=-1)
        out.append((split_heads(k, cfg.n_heads), split_heads(v, cfg.n_heads)))
    return out


# --------------------------------------------------------------------------------------------
# a7/a8: decoder step with KV cache, lm_head, greedy loop
# --------------------------------------------------------------------------------------------
class DecoderState:
    """Self-attention KV cache; rows = segments x channels (a10: channels folded into batch)."""

    def __init__(self, n_rows: int, cfg):
        self.k = [torch.zeros(n_rows, cfg.n_heads, 0, cfg.d_kv) for _ in range(cfg.n_dec_layers)]
        self.v = [torch.zeros(n_rows, cfg.n_heads, 0, cfg.d_kv) for _ in range(cfg.n_dec_layers)]
        self.pos = 0


def dense_ffn(xn: Tensor, W: Dict[str, Tensor], p: str, bf16: bool) -> Tensor:
    ff = _r(torch.relu(xn @ W[p + "wi"].T), bf16)
    return ff @ W[p + "wo2"].T


FP8_MAX = 448.0
# When a list, moe_ffn appends one (rows,) tensor per call: the gap between the k-th and (k+1)-th router logit.
# A row whose gap is below the numerical noise can legitimately route to a different expert on the GPU (a discrete
# change worth ~0.2 in the logits), so parity tests exclude such (row, step) pairs -- the same policy as argmax margins.
MOE_ROUTER_MARGINS = None
# When an iterator, moe_ffn takes its top-k from it instead of from its own router logits: one (rows, k) integer tensor per call (the
# HIP path's recorded choices, ymt3_debug_moe_trace), and appends to MOE_FORCED_DEFICIT, per call, the (rows,) tensor
#   max(0, oracle's k-th largest logit - the smallest oracle logit among the forced experts):
# 0 where the forced set IS the oracle's top-k, otherwise how far below the oracle's cut the worst forced expert lies -- a legitimate
# near-tie choice has a deficit within the numerical noise of the router logits.
MOE_FORCED_SEL = None
MOE_FORCED_DEFICIT = None
MOE_CHOSEN = None          # when a list, moe_ffn appends the (rows, k) experts it used (its own top-k, or the forced ones)


def quant_rows_fp8(x: Tensor):
    """Per-row dynamic e4m3 quantisation: q = fp8(x * (448 / amax)), x ~ q * (amax / 448).  Returns (q as fp32, scale)."""
    amax = x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-12)
    inv = (FP8_MAX / amax).float()
    q = (x * inv).to(torch.float8_e4m3fn).float()
    return q, (amax / FP8_MAX).float()


def moe_ffn(xn: Tensor, W: Dict[str, Tensor], p: str, cfg, bf16: bool) -> Tensor:
    """a11 (build-defined; parity unpinned): router softmax -> top-k -> renormalised expert mix.

    router logits fp32 from the bf16-rounded normed row; top-k by value, ties to the lower expert
    index; gates = softmax over the selected k logits; each expert is a dense ReLU FFN.
    moe_fp8: expert weights are stored as OCP e4m3 with one scale per (expert, matrix); the rows entering
    each expert GEMM are quantised per row (quant_rows_fp8); products accumulate in fp32 and are rescaled
    by row_scale * weight_scale.  The hidden activations are still rounded to bf16 between the two GEMMs.
    """
    E, k, d, dff = cfg.n_experts, cfg.moe_top_k, cfg.d_model, cfg.d_ff
    fp8 = bool(getattr(cfg, "moe_fp8", 0))
    logits = xn @ W[p + "router"].T                                    # (..., E)
    # stable top-k with lowest-index tie-break
    order = torch.argsort(-logits, dim=-1, stable=True)[..., :k]
    if MOE_FORCED_SEL is not None:
        forced = next(MOE_FORCED_SEL).long().reshape(order.shape)
        kth = torch.gather(logits, -1, order)[..., k - 1]
        worst = torch.gather(logits, -1, forced).amin(-1)
        if MOE_FORCED_DEFICIT is not None:
            MOE_FORCED_DEFICIT.append((kth - worst).clamp_min(0).reshape(-1).clone())
        order = forced
    if MOE_CHOSEN is not None:
        MOE_CHOSEN.append(order.reshape(-1, k).clone())
    sel = torch.gather(logits, -1, order)
    gates = torch.softmax(sel, dim=-1)
    if MOE_ROUTER_MARGINS is not None:
        top = torch.topk(logits, k + 1, dim=-1).values
        MOE_ROUTER_MARGINS.append((top[..., k - 1] - top[..., k]).reshape(-1).clone())
    if fp8:
        wi = W[p + "wi_q8"].view(torch.float8_e4m3fn).float().view(E, dff, d)
        wo = W[p + "wo2_q8"].view(torch.float8_e4m3fn).float().view(E, d, dff)
        wi_s, wo_s = W[p + "wi_s"], W[p + "wo2_s"]
    else:
        wi = W[p + "wi"].view(E, dff, d)
        wo = W[p + "wo2"].view(E, d, dff)
    out = torch.zeros_like(xn)
    flat_x = xn.reshape(-1, d)
    flat_o = out.reshape(-1, d)
    flat_order = order.reshape(-1, k)
    flat_g = gates.reshape(-1, k)
    for j in range(k):
        for e in range(E):
            rows = (flat_order[:, j] == e).nonzero().flatten()
            if rows.numel() == 0:
                continue
            if fp8:
                xq, xs = quant_rows_fp8(flat_x[rows])
                hdd = _r(torch.relu((xq @ wi[e].T) * (xs * wi_s[e])), bf16)
                hq, hs = quant_rows_fp8(hdd)
                y = (hq @ wo[e].T) * (hs * wo_s[e])
            else:
                hdd = _r(torch.relu(flat_x[rows] @ wi[e].T), bf16)
                y = hdd @ wo[e].T
            flat_o[rows] += flat_g[rows, j:j + 1] * y
    return flat_o.view_as(xn)


def decoder_step(tokens: Tensor, state: DecoderState, ckv, W: Dict[str, Tensor], cfg, bf16: bool) -> Tensor:
    """One autoregressive step.  tokens: (R,) int64 with R = B*K rows -> logits (R, V) fp32.

    Self-attention appends to the cache (TP: cache_utils.py:144-145) and is causal with the
    unidirectional bucket bias offset by past length (TP: modeling_t5.py:264-279); cross-attention
    reads the per-segment K/V with zero bias; channel c of segment b is row b*K + c.
    """
    K = cfg.n_channels
    R = tokens.shape[0]
    H, inner = cfg.n_heads, cfg.inner
    h = W["dec.embed"][tokens]
    if K > 1:
        h = h + W["dec.chan_embed"][torch.arange(R) % K]
    h = h[:, None, :].float()
    t = state.pos
    bias_d = decoder_bias_by_distance(W["dec.relbias"], t + 1, cfg)           # (H, t+1) by distance
    bias = bias_d[:, torch.arange(t, -1, -1)][None, :, None, :]               # key j -> distance t-j
    for l in range(cfg.n_dec_layers):
        p = f"dec.{l}."
        xn = _r(rmsnorm(h, W[p + "ln1"], cfg.ln_eps), bf16)
        qkv = _r(xn @ W[p + "wqkv"].T, bf16)
        q, k, v = (split_heads(x, H) for x in qkv.split(inner, dim=-1))
        state.k[l] = torch.cat([state.k[l], k], dim=2)
        state.v[l] = torch.cat([state.v[l], v], dim=2)
        a = merge_heads(attention(q, state.k[l], state.v[l], bias, bf16, round_p=False))
        h = h + a @ W[p + "wo"].T
        xn = _r(rmsnorm(h, W[p + "ln2"], cfg.ln_eps), bf16)
        q = split_heads(_r(xn @ W[p + "wq_c"].T, bf16), H)
        kc, vc = ckv[l]
        if K > 1:
            kc = kc.repeat_interleave(K, dim=0)
            vc = vc.repeat_interleave(K, dim=0)
        a = merge_heads(attention(q, kc, vc, None, bf16, round_p=False))
        h = h + a @ W[p + "wo_c"].T
        xn = _r(rmsnorm(h, W[p + "ln3"], cfg.ln_eps), bf16)
        if getattr(cfg, "dec_ffn", 0) == 1:
            h = h + moe_ffn(xn, W, p, cfg, bf16)
        else:
            h = h + dense_ffn(xn, W, p, bf16)
    state.pos += 1
    xn = _r(rmsnorm(h, W["dec.ln_f"], cfg.ln_eps), bf16)
    return (xn @ W["dec.lm_head"].T)[:, 0, :]


def greedy_decode(enc_out: Tensor, W: Dict[str, Tensor], cfg, n_steps: int, bf16: bool,
                  forced: Optional[Tensor] = None, return_logits: bool = False):
    """Greedy loop (TP: generation/utils.py:2876-2941).  Returns tokens (B, K, n_steps) int32.

    Step 0 consumes decoder_start = pad_id; argmax over fp32 logits (first index on ties);
    once a row has produced eos_id every later token is pad_id (TP: utils.py:2928-2929).
    `forced` (B, K, n_steps) teacher-forces the fed-back tokens (the emitted tokens are still the
    argmax) so logits can be compared step by step without trajectory divergence.
    """
    B = enc_out.shape[0]
    K = cfg.n_channels
    R = B * K
    ckv = cross_kv(enc_out, W, cfg, bf16)
    state = DecoderState(R, cfg)
    cur = torch.full((R,), cfg.pad_id, dtype=torch.long)
    finished = torch.zeros(R, dtype=torch.bool)
    out = torch.zeros(R, n_steps, dtype=torch.int32)
    all_logits = []
    for t in range(n_steps):
        logits = decoder_step(cur, state, ckv, W, cfg, bf16)
        nxt = torch.argmax(logits.float(), dim=-1)
        if cfg.eos_id >= 0:
            nxt = torch.where(finished, torch.full_like(nxt, cfg.pad_id), nxt)
            finished = finished | (nxt == cfg.eos_id)
        out[:, t] = nxt.to(torch.int32)
        if return_logits:
            all_logits.append(logits.clone())
        cur = forced.reshape(R, -1)[:, t].long() if forced is not None else nxt
    toks = out.view(B, K, n_steps)
    if return_logits:
        return toks, torch.stack(all_logits, 1).view(B, K, n_steps, -1)
    return toks


def transcribe_segments(audio: Tensor, W: Dict[str, Tensor], cfg, n_steps: Optional[int] = None,
                        bf16: bool = True) -> Tensor:
    """audio (B, S) -> token ids (B, K, L): the whole hot path on the CPU."""
    _, enc = encode(audio, W, cfg, bf16)
    return greedy_decode(enc, W, cfg, n_steps or cfg.max_decode_len, bf16)


def synthetic_audio(B: int, cfg, seed: int = 0) -> Tensor:
    """SURVEY section 8c/8d synthetic input: N(0, 0.1^2) clipped to +-1 plus a 440 Hz + 880 Hz tone mix."""
    g = torch.Generator().manual_seed(seed)
    S = cfg.segment_samples
    noise = (0.1 * torch.randn(B, S, generator=g)).clamp(-1, 1)
    t = torch.arange(S, dtype=torch.float64) / cfg.sample_rate
    f0 = 440.0 * (1.0 + 0.25 * torch.arange(B, dtype=torch.float64))[:, None]
    tone = 0.3 * torch.sin(2 * math.pi * f0 * t) + 0.15 * torch.sin(2 * math.pi * 2 * f0 * t)
    return (noise + tone.float()).clamp(-1, 1).contiguous()
