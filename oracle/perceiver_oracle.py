"""CPU oracle of the latent-array (Perceiver) encoder, a9 of SURVEY.md section 8.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference tree has no code, and the only Perceiver in the container is the generic HF one
(TP: transformers/models/perceiver/modeling_perceiver.py:136-243 attention with pre-LN on queries and on
keys/values, :328-329 query residual, :335-351 MLP, :418-525 cross-attention followed by self-attention
blocks).  This build keeps that STRUCTURE -- a learned latent array cross-attends once to the projected
frames, then latent self-attention blocks -- but with the T5 numerics of the rest of the path (RMS norm,
no 1/sqrt(d) scale, ReLU FFN, bf16 rounding points of DESIGN.md section 2), so that the same kernels serve
both encoders.  YourMT3+'s spectral/temporal factorisation ("TF") is not reproduced: nothing available
offline specifies it (SURVEY section 2.2 star-4).
"""
from __future__ import annotations

from typing import Dict

import torch

from oracle.ymt3_oracle import _r, attention, encoder_t5, merge_heads, rmsnorm, split_heads

Tensor = torch.Tensor


def latent_cross_attention(h: Tensor, W: Dict[str, Tensor], cfg, bf16: bool) -> Tensor:
    """frames h (B, T, d) fp32 -> latent residual stream z (B, n_latents, d) fp32."""
    B = h.shape[0]
    H, inner = cfg.n_heads, cfg.inner
    f = _r(rmsnorm(h, W["ptf.ca.ln_kv"], cfg.ln_eps), bf16)
    kv = _r(f @ W["ptf.ca.wkv"].T, bf16)
    k, v = (split_heads(t, H) for t in kv.split(inner, dim=-1))
    z = W["ptf.latents"][None].expand(B, -1, -1).float().contiguous()
    q = split_heads(_r(_r(rmsnorm(z, W["ptf.ca.ln_q"], cfg.ln_eps), bf16) @ W["ptf.ca.wq"].T, bf16), H)
    a = merge_heads(attention(q, k, v, None, bf16, round_p=True))
    z = z + a @ W["ptf.ca.wo"].T
    xn = _r(rmsnorm(z, W["ptf.ca.ln_ff"], cfg.ln_eps), bf16)
    ff = _r(torch.relu(xn @ W["ptf.ca.wi"].T), bf16)
    return z + ff @ W["ptf.ca.wo2"].T


def encoder_perceiver_tf(h: Tensor, W: Dict[str, Tensor], cfg, bf16: bool) -> Tensor:
    assert cfg.n_latents == h.shape[1], "this build ties the latent array length to the frame count"
    z = latent_cross_attention(h, W, cfg, bf16)
    return encoder_t5(z, W, cfg, bf16)          # latent self-attention blocks + final norm
