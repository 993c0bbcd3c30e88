"""Host-side derived tables the HIP library expects inside the weight blob.

Pure integer / float64 host work, done once per (config, checkpoint):
  * Hann window, FFT twiddles, real-FFT untangle factors, CSR mel filterbank (a1/a2)
  * relative-position bucket -> bias tables indexed by offset (encoder) / distance (decoder) (a5)

The bucket function restates TP: transformers/models/t5/modeling_t5.py:216-262 (fp32 log, truncation
toward zero) so that boundaries (e.g. |rel| = 16, 32, 64, 128) land in the same bucket as HF T5;
tests/test_cpu_host.py::test_bucket_tables_match_oracle_hf_and_golden checks it bit-exactly against that code and against the oracle.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

from .config import YMT3Config


def relative_position_bucket(rel: np.ndarray, bidirectional: bool, num_buckets: int, max_distance: int) -> np.ndarray:
    """rel = key_pos - query_pos (any int array) -> bucket ids in [0, num_buckets)."""
    rel = np.asarray(rel, dtype=np.int64)
    ret = np.zeros_like(rel)
    n = num_buckets
    if bidirectional:
        n //= 2
        ret = ret + (rel > 0).astype(np.int64) * n
        dist = np.abs(rel)
    else:
        dist = np.maximum(-rel, 0)
    max_exact = n // 2
    safe = np.maximum(dist, 1).astype(np.float32)
    val = np.log(safe / np.float32(max_exact)).astype(np.float32)
    val = (val / np.float32(math.log(max_distance / max_exact))).astype(np.float32) * np.float32(n - max_exact)
    large = np.minimum(max_exact + val.astype(np.int64), n - 1)
    return ret + np.where(dist < max_exact, dist, large)


def encoder_bias_table(relbias: torch.Tensor, T: int, cfg: YMT3Config) -> torch.Tensor:
    """(H, 2T-1) fp32, column = key - query + (T-1)."""
    b = relative_position_bucket(np.arange(-(T - 1), T), True, cfg.rel_buckets, cfg.rel_max_distance)
    return relbias.float()[torch.from_numpy(b)].T.contiguous()


def decoder_bias_table(relbias: torch.Tensor, L: int, cfg: YMT3Config) -> torch.Tensor:
    """(H, L) fp32, column = query - key >= 0 (causal distance)."""
    b = relative_position_bucket(-np.arange(L), False, cfg.rel_buckets, cfg.rel_max_distance)
    return relbias.float()[torch.from_numpy(b)].T.contiguous()


def mel_filterbank(cfg: YMT3Config) -> np.ndarray:
    """(n_mels, n_freqs) float64 HTK triangles, unnormalised: m = 2595 log10(1 + f / 700)."""
    n_freqs = cfg.n_freqs
    freqs = np.linspace(0.0, cfg.sample_rate / 2.0, n_freqs)
    lo = 2595.0 * np.log10(1.0 + cfg.f_min / 700.0)
    hi = 2595.0 * np.log10(1.0 + cfg.f_max / 700.0)
    pts = 700.0 * (10.0 ** (np.linspace(lo, hi, cfg.n_mels + 2) / 2595.0) - 1.0)
    width = np.diff(pts)
    rise = (freqs[None, :] - pts[:-2, None]) / width[:-1, None]
    fall = (pts[2:, None] - freqs[None, :]) / width[1:, None]
    return np.clip(np.minimum(rise, fall), 0.0, None)


def frontend_tables(cfg: YMT3Config) -> Dict[str, torch.Tensor]:
    n_fft, n2 = cfg.n_fft, cfg.n_fft // 2
    n = np.arange(n_fft, dtype=np.float64)
    window = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)
    m = np.arange(n2, dtype=np.float64)
    tw = np.stack([np.cos(2 * np.pi * m / n2), -np.sin(2 * np.pi * m / n2)], -1)
    k = np.arange(n2 + 1, dtype=np.float64)
    untw = np.stack([np.cos(2 * np.pi * k / n_fft), -np.sin(2 * np.pi * k / n_fft)], -1)
    fb = mel_filterbank(cfg).astype(np.float32)
    start, length, off, w = [], [], [], []
    for r in range(cfg.n_mels):
        nz = np.nonzero(fb[r])[0]
        s, e = (int(nz[0]), int(nz[-1]) + 1) if nz.size else (0, 0)
        start.append(s); length.append(e - s); off.append(len(w))
        w.extend(fb[r, s:e].tolist())
    if not w:
        w = [0.0]
    return {
        "fe.window": torch.from_numpy(window.astype(np.float32)),
        "fe.tw": torch.from_numpy(tw.astype(np.float32)),
        "fe.untw": torch.from_numpy(untw.astype(np.float32)),
        "fe.mel_start": torch.tensor(start, dtype=torch.int32),
        "fe.mel_len": torch.tensor(length, dtype=torch.int32),
        "fe.mel_off": torch.tensor(off, dtype=torch.int32),
        "fe.mel_w": torch.tensor(w, dtype=torch.float32),
    }


def derived_tables(W: Dict[str, torch.Tensor], cfg: YMT3Config) -> Dict[str, torch.Tensor]:
    out = frontend_tables(cfg)
    out["enc.bias_off"] = encoder_bias_table(W["enc.relbias"], cfg.n_frames, cfg)
    out["dec.bias_dist"] = decoder_bias_table(W["dec.relbias"], cfg.max_decode_len, cfg)
    if "ptf.relbias" in W:                  # temporal transformer of the Perceiver-TF encoder: same bucket function, its own table
        out["ptf.bias_off"] = encoder_bias_table(W["ptf.relbias"], cfg.n_frames, cfg)
    return out
