"""Shape/behaviour specification of the transcription hot path.

The reference checkout (/root/reference) holds only README.md and LICENSE, so there
is no upstream config file to mirror.  The values below are this
build's own written spec; defaults follow SURVEY.md section 8 ("Shapes use: ...") and
BASELINE.json `configs`.  The same dataclass is consumed by the CPU oracle
(oracle/ymt3_oracle.py), the weight-blob writer (yourmt3_amd/weights.py) and -- packed
as the C struct `ymt3_config` of include/ymt3.h -- by the HIP library.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, asdict, replace

PAD_ID = 0
EOS_ID = 1
UNK_ID = 2

ENC_T5 = 0
ENC_PERCEIVER_TF = 1
FFN_DENSE = 0
FFN_MOE = 1


@dataclass(frozen=True)
class YMT3Config:
    # --- audio front-end (a1, a2) ---
    sample_rate: int = 16000
    segment_samples: int = 32767      # 2.048 s -> 1 + S // hop = 256 frames (centre padded)
    n_fft: int = 2048
    hop: int = 128
    n_mels: int = 128
    f_min: float = 50.0
    f_max: float = 8000.0
    log_floor: float = 1e-8
    # --- transformer dims (HF t5-small dims) ---
    d_model: int = 512
    d_ff: int = 2048
    n_heads: int = 8
    d_kv: int = 64
    n_enc_layers: int = 6
    n_dec_layers: int = 6
    vocab: int = 1536
    rel_buckets: int = 32
    rel_max_distance: int = 128
    ln_eps: float = 1e-6
    # --- decoder behaviour ---
    max_decode_len: int = 1024        # L: tokens emitted per (segment, channel)
    n_channels: int = 1               # K: 13 for the multi-track decoder (config 4)
    eos_id: int = EOS_ID              # -1 disables the EOS->PAD fill (forced-length bench)
    pad_id: int = PAD_ID
    # --- architecture variants ---
    encoder_type: int = ENC_T5        # ENC_PERCEIVER_TF for config 3
    n_latents: int = 32               # Perceiver-TF: latents per frame (32 or 64; independent of the frame count)
    dec_ffn: int = FFN_DENSE          # FFN_MOE for config 5
    n_experts: int = 8
    moe_top_k: int = 2
    moe_fp8: int = 0                  # 1: expert GEMMs on OCP e4m3 MFMA (per-expert weight scale, per-row activation scale)
    # --- Perceiver-TF encoder (oracle/perceiver_oracle.py) ---
    ptf_d: int = 128                  # latent / spectral-token width (heads of 64)
    ptf_blocks: int = 3               # [spectral cross-attention, latent transformer, temporal transformer] x blocks
    ptf_dff: int = 512                # FFN width inside the three sub-layers

    @property
    def n_frames(self) -> int:
        return 1 + self.segment_samples // self.hop

    @property
    def n_freqs(self) -> int:
        return self.n_fft // 2 + 1

    @property
    def inner(self) -> int:
        return self.n_heads * self.d_kv

    @property
    def segment_seconds(self) -> float:
        return (self.segment_samples + 1) / self.sample_rate

    def with_(self, **kw) -> "YMT3Config":
        return replace(self, **kw)

    def to_dict(self) -> dict:
        return asdict(self)


class CConfig(ctypes.Structure):
    """Mirror of `struct ymt3_config` in include/ymt3.h (field order is ABI)."""
    _fields_ = [
        ("sample_rate", ctypes.c_int32), ("segment_samples", ctypes.c_int32),
        ("n_fft", ctypes.c_int32), ("hop", ctypes.c_int32), ("n_mels", ctypes.c_int32),
        ("f_min", ctypes.c_float), ("f_max", ctypes.c_float), ("log_floor", ctypes.c_float),
        ("d_model", ctypes.c_int32), ("d_ff", ctypes.c_int32), ("n_heads", ctypes.c_int32),
        ("d_kv", ctypes.c_int32), ("n_enc_layers", ctypes.c_int32), ("n_dec_layers", ctypes.c_int32),
        ("vocab", ctypes.c_int32), ("rel_buckets", ctypes.c_int32), ("rel_max_distance", ctypes.c_int32),
        ("ln_eps", ctypes.c_float),
        ("max_decode_len", ctypes.c_int32), ("n_channels", ctypes.c_int32),
        ("eos_id", ctypes.c_int32), ("pad_id", ctypes.c_int32),
        ("encoder_type", ctypes.c_int32), ("n_latents", ctypes.c_int32),
        ("dec_ffn", ctypes.c_int32), ("n_experts", ctypes.c_int32), ("moe_top_k", ctypes.c_int32), ("moe_fp8", ctypes.c_int32),
        ("ptf_d", ctypes.c_int32), ("ptf_blocks", ctypes.c_int32), ("ptf_dff", ctypes.c_int32),
        ("max_batch", ctypes.c_int32),
    ]


def to_c(cfg: YMT3Config, max_batch: int) -> CConfig:
    c = CConfig()
    for name, _ in CConfig._fields_:
        if name == "max_batch":
            c.max_batch = int(max_batch)
        else:
            setattr(c, name, getattr(cfg, name))
    return c


# BASELINE.json `configs`, by index
def baseline_config(i: int) -> YMT3Config:
    base = YMT3Config()
    if i == 0:      # single 2 s segment, CPU reference path
        return base
    if i == 1:      # MT3 base (T5-small) bf16, batch 64, 1024-token decoder
        return base.with_(eos_id=-1)
    if i == 2:      # Perceiver-TF encoder + T5 decoder, batch 256
        return base.with_(encoder_type=ENC_PERCEIVER_TF, n_latents=32, n_enc_layers=0, eos_id=-1)
    if i == 3:      # 13-channel multi-track decoder, 256 tokens per channel
        return base.with_(n_channels=13, max_decode_len=256, eos_id=-1)
    if i == 4:      # MoE decoder FFN (8 experts)
        return base.with_(dec_ffn=FFN_MOE, moe_fp8=1, eos_id=-1)
    raise IndexError(i)
