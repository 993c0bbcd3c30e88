"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the audio ingest step that precedes the hot path
(SURVEY.md section 8f rank 2): interleaved PCM -> mono -> polyphase FIR resample to 16 kHz -> zero-padded
fixed-length segments.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

PARITY UNPINNED against the reference (its tree holds no code).  The resampler restates
the published algorithm of a third-party dependency the upstream project is believed to use through its audio
loader, scipy.signal.resample_poly (TP: scipy/signal/_signaltools.py `resample_poly`, scipy 1.15.3: Kaiser(5.0)
windowed-sinc low-pass of half length 10*max(up, down), cutoff 1/max(up, down), gain `up`, zero extension, output
sample n aligned with input time n*down/up); tests/test_ingest.py pins it to scipy's own output to 1e-9.
"""
from __future__ import annotations

from math import gcd

import numpy as np


def rates(sr_in: int, sr_out: int):
    g = gcd(int(sr_in), int(sr_out))
    return int(sr_out) // g, int(sr_in) // g          # up, down


def lowpass_taps(up: int, down: int) -> np.ndarray:
    """The FIR of TP: resample_poly (firwin(2*half_len+1, 1/max_rate, window=('kaiser', 5.0)) * up), float64."""
    if up == 1 and down == 1:
        return np.ones(1)
    max_rate = max(up, down)
    half_len = 10 * max_rate
    n = 2 * half_len + 1
    m = np.arange(n, dtype=np.float64) - half_len
    fc = 1.0 / max_rate
    h = fc * np.sinc(fc * m)
    # symmetric Kaiser window, beta = 5
    w = np.i0(5.0 * np.sqrt(np.maximum(0.0, 1.0 - (m / half_len) ** 2))) / np.i0(5.0)
    h = h * w
    h = h / h.sum()
    return h * up


def plan(n_in: int, up: int, down: int):
    """(n_out, n_pre_remove, padded filter) exactly as TP: resample_poly lays them out for upfirdn."""
    h = lowpass_taps(up, down)
    n_out = (n_in * up + down - 1) // down
    if up == 1 and down == 1:
        return n_out, 0, h
    half_len = (len(h) - 1) // 2
    n_pre_pad = down - half_len % down
    n_pre_remove = (half_len + n_pre_pad) // down
    hp = np.concatenate([np.zeros(n_pre_pad), h])
    return n_out, n_pre_remove, hp


def polyphase_table(hp: np.ndarray, up: int):
    """P[phase][j] = hp[phase + j*up]; rows padded to a multiple of 4 taps."""
    J = -(-len(hp) // up)
    Jp = (J + 3) // 4 * 4
    P = np.zeros((up, Jp), dtype=np.float64)
    for ph in range(up):
        col = hp[ph::up]
        P[ph, :len(col)] = col
    return P, J


def mono(pcm: np.ndarray) -> np.ndarray:
    """(n_frames, n_channels) int16 or float32 -> (n_frames,) float32 in [-1, 1)."""
    pcm = np.asarray(pcm)
    if pcm.ndim == 1:
        pcm = pcm[:, None]
    if pcm.dtype == np.int16:
        x = pcm.astype(np.float32) * np.float32(1.0 / 32768.0)
    elif pcm.dtype == np.float32:
        x = pcm
    else:
        raise ValueError("pcm must be int16 or float32")
    return (x.sum(axis=1, dtype=np.float32) * np.float32(1.0 / pcm.shape[1])).astype(np.float32)


def resample(x: np.ndarray, sr_in: int, sr_out: int) -> np.ndarray:
    """y[n] = sum_j hp[((n + r) * down) % up + j * up] * x[((n + r) * down) // up - j], zero outside x; float32 taps and
    samples, float64 accumulation (the kernel accumulates in float32: tolerance in tests/test_gpu_parity.py)."""
    up, down = rates(sr_in, sr_out)
    n_in = x.shape[0]
    n_out, r, hp = plan(n_in, up, down)
    P, J = polyphase_table(hp, up)
    P32 = P.astype(np.float32).astype(np.float64)
    xs = np.concatenate([np.zeros(P.shape[1], dtype=np.float64), x.astype(np.float64), np.zeros(P.shape[1] + 2, dtype=np.float64)])
    q = (np.arange(n_out, dtype=np.int64) + r) * down
    ph, k0 = q % up, q // up
    y = np.zeros(n_out, dtype=np.float64)
    off = P.shape[1]
    for j in range(J):
        idx = k0 - j
        ok = (idx >= -off) & (idx < n_in + off)
        y += np.where(ok, P32[ph, j] * xs[np.clip(idx + off, 0, len(xs) - 1)], 0.0)
    return y.astype(np.float32)


def n_segments(n_in: int, sr_in: int, sr_out: int, segment_samples: int) -> int:
    up, down = rates(sr_in, sr_out)
    n_out = (n_in * up + down - 1) // down
    return max(1, -(-n_out // segment_samples))


def ingest(pcm: np.ndarray, sr_in: int, sr_out: int, segment_samples: int) -> np.ndarray:
    """-> (n_seg, segment_samples) float32, the last segment zero padded."""
    y = resample(mono(pcm), sr_in, sr_out)
    n_seg = max(1, -(-y.shape[0] // segment_samples))
    out = np.zeros(n_seg * segment_samples, dtype=np.float32)
    out[:y.shape[0]] = y
    return out.reshape(n_seg, segment_samples)
