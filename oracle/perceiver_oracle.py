"""CPU oracle of the Perceiver-TF encoder, a9 of SURVEY.md section 8.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference tree has no code, and the only Perceiver in the container is the generic HF one
(TP: transformers/models/perceiver/modeling_perceiver.py:136-243 attention with pre-LN on queries and on
keys/values, :328-329 query residual, :335-351 MLP, :418-525 cross-attention followed by self-attention).
SURVEY section 8 row a9 gives the SHAPE of YourMT3+'s encoder -- (B, T, F', C) spectral features, a latent array that
cross-attends to the spectral axis, latent self-attention, temporal self-attention, (B, T, n_latents, d) out -- and
nothing offline gives more than that.  This file is therefore the build's own written spec of that shape (DESIGN.md
section 8), with the numerics of the rest of the path (RMS norm, no 1/sqrt(d) scale, ReLU FFN, bf16 rounding points of
DESIGN.md section 2) so that the encoder shares the GEMM / norm / attention kernels of the T5 encoder:

  x[b,t,f,:]  = mel[b,t,f] * spec_w + spec_pos[f]                 spectral features (B, T, F' = n_mels, C = ptf_d), fp32
  z[b,t,k,:]  = latents[k]                                        latent residual stream (B, T, K = n_latents, D = ptf_d)
  per block:    spectral cross-attention   for every (b, t):  K latent queries over the F' spectral tokens of frame t
                latent transformer         for every (b, t):  self-attention among the K latents
                temporal transformer       for every (b, k):  self-attention over the T frames, T5 relative position bias
                (each: pre-norm, ptf_d / 64 heads of 64, residual, then a pre-norm ReLU FFN of width ptf_dff)
  out         = (B, T, K, D) -> rmsnorm per latent -> concat over k -> linear (K*D -> d_model) -> final rmsnorm (B, T, d_model)

`n_latents` is free of the frame count; the decoder still cross-attends over T keys per segment.
"""
from __future__ import annotations

from typing import Dict

import torch

from oracle.ymt3_oracle import _r, attention, encoder_bias_by_offset, merge_heads, rmsnorm, split_heads

Tensor = torch.Tensor


def _mha(q: Tensor, kv_k: Tensor, kv_v: Tensor, bias, bf16: bool) -> Tensor:
    """(N, Tq, D), (N, Tk, D) x2 -> (N, Tq, D): heads of 64, softmax(q k^T + bias) v, P rounded to bf16 as the MFMA kernel does."""
    H = q.shape[-1] // 64
    return merge_heads(attention(split_heads(q, H), split_heads(kv_k, H), split_heads(kv_v, H), bias, bf16, round_p=True))


def _ffn(z: Tensor, W: Dict[str, Tensor], p: str, eps: float, bf16: bool) -> Tensor:
    xn = _r(rmsnorm(z, W[p + "ln_ff"], eps), bf16)
    ff = _r(torch.relu(xn @ W[p + "wi"].T), bf16)
    return z + ff @ W[p + "wo2"].T


def spectral_features(mel: Tensor, W: Dict[str, Tensor], cfg, bf16: bool) -> Tensor:
    """(B, T, F') log-mel -> normed spectral tokens (B, T, F', C), rounded to bf16 (the K/V GEMM operand)."""
    x = mel[..., None] * W["ptf.spec_w"] + W["ptf.spec_pos"]
    return _r(rmsnorm(x, W["ptf.ln_x"], cfg.ln_eps), bf16)


def perceiver_tf_latents(mel: Tensor, W: Dict[str, Tensor], cfg, bf16: bool) -> Tensor:
    """(B, T, F') -> the latent residual stream after the last block, (B, T, K, D) fp32."""
    B, T, F = mel.shape
    K, D, eps = cfg.n_latents, cfg.ptf_d, cfg.ln_eps
    xn = spectral_features(mel, W, cfg, bf16).reshape(B * T, F, D)
    z = W["ptf.latents"][None, None].expand(B, T, K, D).to(mel.dtype).contiguous()
    H = D // 64
    bias_off = encoder_bias_by_offset(W["ptf.relbias"], T, cfg)                # (H, 2T-1)
    idx = (torch.arange(T)[None, :] - torch.arange(T)[:, None]) + (T - 1)        # key - query + T-1
    tbias = bias_off[:, idx][None]                                               # (1, H, T, T)
    assert bias_off.shape[0] == H
    for blk in range(cfg.ptf_blocks):
        p = f"ptf.{blk}."
        # spectral cross-attention: sequences are the (b, t) pairs
        kv = _r(xn @ W[p + "s.wkv"].T, bf16)
        zq = z.reshape(B * T, K, D)
        q = _r(_r(rmsnorm(zq, W[p + "s.ln_q"], eps), bf16) @ W[p + "s.wq"].T, bf16)
        a = _mha(q, kv[..., :D], kv[..., D:], None, bf16)
        zq = zq + a @ W[p + "s.wo"].T
        zq = _ffn(zq, W, p + "s.", eps, bf16)
        # latent transformer: the same sequences, keys = the K latents
        qkv = _r(_r(rmsnorm(zq, W[p + "l.ln1"], eps), bf16) @ W[p + "l.wqkv"].T, bf16)
        a = _mha(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], None, bf16)
        zq = zq + a @ W[p + "l.wo"].T
        zq = _ffn(zq, W, p + "l.", eps, bf16)
        # temporal transformer: sequences are the (b, k) pairs, positions the T frames
        zt = zq.reshape(B, T, K, D).transpose(1, 2).reshape(B * K, T, D)
        qkv = _r(_r(rmsnorm(zt, W[p + "t.ln1"], eps), bf16) @ W[p + "t.wqkv"].T, bf16)
        a = _mha(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], tbias, bf16)
        zt = zt + a @ W[p + "t.wo"].T
        zt = _ffn(zt, W, p + "t.", eps, bf16)
        z = zt.reshape(B, K, T, D).transpose(1, 2).contiguous()
    return z


def in_double(W: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """The weights cast to float64: with a float64 `mel` the encoder below accumulates every sum in double, rounding points unchanged."""
    return {k: (v.double() if v.is_floating_point() else v) for k, v in W.items()}


def encoder_perceiver_tf(mel: Tensor, W: Dict[str, Tensor], cfg, bf16: bool) -> Tensor:
    """(B, T, F') log-mel -> (B, T, d_model) encoder output the decoder cross-attends to."""
    B, T, _ = mel.shape
    z = perceiver_tf_latents(mel, W, cfg, bf16)
    zn = _r(rmsnorm(z, W["ptf.ln_out"], cfg.ln_eps), bf16).reshape(B, T, cfg.n_latents * cfg.ptf_d)
    y = zn @ W["ptf.out_w"].T
    return _r(rmsnorm(y, W["enc.ln_f"], cfg.ln_eps), bf16)
