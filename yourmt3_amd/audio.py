"""Host side of the audio ingest: WAV file IO with the stdlib `wave` reader (torchaudio /
soundfile absent) and synthetic test audio.  Mixing, resampling and slicing are NOT done here: they run on the device
(`YourMT3.ingest` -> C ABI `ymt3_ingest`); `slice_padded_array` only reshapes an array that is already mono at the model rate."""
from __future__ import annotations

import wave
from typing import Tuple

import numpy as np


def load_wav_pcm(path: str) -> Tuple[np.ndarray, int]:
    """WAV file -> ((n_frames, n_channels) int16 or float32 PCM exactly as stored, sample rate): the host does file
    IO only; mixing, resampling and slicing happen on the device (YourMT3.ingest)."""
    with wave.open(path, "rb") as w:
        sr, nch, width, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2")
    elif width == 4:
        x = (np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    elif width == 1:
        x = ((np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0).astype(np.float32)
    else:
        raise ValueError(f"unsupported sample width {width}")
    return x.reshape(-1, nch), sr


def slice_padded_array(x: np.ndarray, slice_length: int, pad: bool = True) -> np.ndarray:
    """(n,) -> (n_seg, 1, slice_length): consecutive non-overlapping slices, the last one zero padded."""
    n = x.shape[0]
    n_seg = max(1, -(-n // slice_length)) if pad else n // slice_length
    out = np.zeros((n_seg, 1, slice_length), dtype=np.float32)
    for i in range(n_seg):
        seg = x[i * slice_length:(i + 1) * slice_length]
        out[i, 0, :seg.shape[0]] = seg
    return out


def synthetic_segments(n: int, segment_samples: int, sample_rate: int = 16000, seed: int = 0) -> np.ndarray:
    """(n, segment_samples) float32 test audio: N(0, 0.1^2) clipped to +-1 plus a tone pair per segment
    (SURVEY.md section 8d synthetic input); used by bench.py and the measurement scripts."""
    rng = np.random.default_rng(seed)
    noise = np.clip(0.1 * rng.standard_normal((n, segment_samples)), -1, 1)
    t = np.arange(segment_samples, dtype=np.float64) / sample_rate
    f0 = 220.0 * (1.0 + 0.03 * np.arange(n, dtype=np.float64))[:, None]
    tone = 0.3 * np.sin(2 * np.pi * f0 * t) + 0.15 * np.sin(4 * np.pi * f0 * t)
    return np.clip(noise + tone, -1, 1).astype(np.float32)
