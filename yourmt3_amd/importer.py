"""Weight importer: a T5-style state_dict (HF `T5ForConditionalGeneration` naming, which upstream YourMT3's
`t5mod.py` builds on per SURVEY.md section 9, UNVERIFIED) -> this build's named tensors / blob.

No real checkpoint is available offline, so the importer is exercised on randomly initialised HF T5 modules
(tests/test_importer.py): the imported weights must reproduce HF's own encoder output and greedy ids
through the oracle, and (GPU test) through the HIP path.  Everything the T5 part of the path needs is
mapped; the audio front-end projection (`in_proj`) has no T5 counterpart and is passed separately.
"""
from __future__ import annotations

from typing import Dict, Mapping, Optional

import torch

from .config import YMT3Config


def _bf16(t: torch.Tensor) -> torch.Tensor:
    return t.detach().float().to(torch.bfloat16).float()


def from_t5_state_dict(sd: Mapping[str, torch.Tensor], cfg: YMT3Config, in_proj_w: Optional[torch.Tensor] = None,
                       in_proj_b: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Map HF-T5-named tensors to this build's names; matrices are rounded to bf16 (the blob's storage type)."""
    W: Dict[str, torch.Tensor] = {}
    d = cfg.d_model
    W["in_proj.w"] = _bf16(in_proj_w) if in_proj_w is not None else torch.zeros(d, cfg.n_mels)
    W["in_proj.b"] = in_proj_b.detach().float() if in_proj_b is not None else torch.zeros(d)
    for l in range(cfg.n_enc_layers):
        h, p = f"encoder.block.{l}.layer.", f"enc.{l}."
        W[p + "ln1"] = sd[h + "0.layer_norm.weight"].detach().float()
        W[p + "wqkv"] = _bf16(torch.cat([sd[h + f"0.SelfAttention.{n}.weight"] for n in "qkv"], 0))
        W[p + "wo"] = _bf16(sd[h + "0.SelfAttention.o.weight"])
        W[p + "ln2"] = sd[h + "1.layer_norm.weight"].detach().float()
        W[p + "wi"] = _bf16(sd[h + "1.DenseReluDense.wi.weight"])
        W[p + "wo2"] = _bf16(sd[h + "1.DenseReluDense.wo.weight"])
    W["enc.relbias"] = sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"].detach().float()
    W["enc.ln_f"] = sd["encoder.final_layer_norm.weight"].detach().float()
    emb = sd.get("decoder.embed_tokens.weight", sd.get("shared.weight"))
    W["dec.embed"] = _bf16(emb)
    for l in range(cfg.n_dec_layers):
        h, p = f"decoder.block.{l}.layer.", f"dec.{l}."
        W[p + "ln1"] = sd[h + "0.layer_norm.weight"].detach().float()
        W[p + "wqkv"] = _bf16(torch.cat([sd[h + f"0.SelfAttention.{n}.weight"] for n in "qkv"], 0))
        W[p + "wo"] = _bf16(sd[h + "0.SelfAttention.o.weight"])
        W[p + "ln2"] = sd[h + "1.layer_norm.weight"].detach().float()
        W[p + "wq_c"] = _bf16(sd[h + "1.EncDecAttention.q.weight"])
        W[p + "wkv_c"] = _bf16(torch.cat([sd[h + "1.EncDecAttention.k.weight"], sd[h + "1.EncDecAttention.v.weight"]], 0))
        W[p + "wo_c"] = _bf16(sd[h + "1.EncDecAttention.o.weight"])
        W[p + "ln3"] = sd[h + "2.layer_norm.weight"].detach().float()
        W[p + "wi"] = _bf16(sd[h + "2.DenseReluDense.wi.weight"])
        W[p + "wo2"] = _bf16(sd[h + "2.DenseReluDense.wo.weight"])
    W["dec.relbias"] = sd["decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"].detach().float()
    W["dec.ln_f"] = sd["decoder.final_layer_norm.weight"].detach().float()
    W["dec.lm_head"] = _bf16(sd["lm_head.weight"])
    for name, t in W.items():
        if not torch.isfinite(t).all():
            raise ValueError(f"non-finite values in imported tensor {name}")
    _check_shapes(W, cfg)
    return W


def _check_shapes(W: Dict[str, torch.Tensor], cfg: YMT3Config) -> None:
    d, inner = cfg.d_model, cfg.inner
    want = {"dec.embed": (cfg.vocab, d), "dec.lm_head": (cfg.vocab, d), "enc.relbias": (cfg.rel_buckets, cfg.n_heads),
            "dec.relbias": (cfg.rel_buckets, cfg.n_heads), "enc.0.wqkv": (3 * inner, d), "dec.0.wkv_c": (2 * inner, d),
            "enc.0.wi": (cfg.d_ff, d), "dec.0.wo2": (d, cfg.d_ff)}
    for k, shp in want.items():
        if tuple(W[k].shape) != shp:
            raise ValueError(f"{k}: checkpoint shape {tuple(W[k].shape)} does not match the config {shp}")
