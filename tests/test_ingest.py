"""CPU tests of the audio-ingest oracle: the polyphase restatement in
oracle/ingest_oracle.py against the third-party routine it restates (scipy.signal.resample_poly, scipy 1.15.3),
and the segment arithmetic the C ABI's ymt3_ingest_plan mirrors.  The reference tree holds no audio code or
fixtures: parity with the reference itself is unpinned."""
import numpy as np
import pytest
from scipy.signal import resample_poly

from oracle import ingest_oracle as io


@pytest.mark.parametrize("sr,n", [(44100, 50000), (48000, 30001), (8000, 9000), (22050, 20000), (16000, 5000),
                                  (32000, 7777), (11025, 4000), (44100, 1), (96000, 100)])
def test_polyphase_restatement_matches_scipy(sr, n):
    rng = np.random.default_rng(sr + n)
    x = rng.standard_normal(n)
    up, down = io.rates(sr, 16000)
    ref = resample_poly(x, up, down)
    n_out, r, hp = io.plan(n, up, down)
    assert n_out == ref.shape[0]
    P, J = io.polyphase_table(hp, up)
    off = P.shape[1]
    xs = np.concatenate([np.zeros(off), x, np.zeros(off + 2)])
    q = (np.arange(n_out, dtype=np.int64) + r) * down
    ph, k0 = q % up, q // up
    y = np.zeros(n_out)
    for j in range(J):
        idx = k0 - j
        ok = (idx >= -off) & (idx < n + off)
        y += np.where(ok, P[ph, j] * xs[np.clip(idx + off, 0, len(xs) - 1)], 0.0)
    assert np.abs(y - ref).max() < 1e-9                       # float64 taps: the algorithm itself
    y32 = io.resample(x.astype(np.float32), sr, 16000)        # float32 taps and samples: what the kernel is held to
    assert y32.dtype == np.float32 and np.abs(y32 - ref).max() < 2e-6 * max(1.0, np.abs(x).max())


def test_mono_mix_and_segment_layout():
    rng = np.random.default_rng(0)
    pcm = rng.integers(-32768, 32767, size=(44100, 2), dtype=np.int16)
    m = io.mono(pcm)
    assert m.dtype == np.float32 and np.allclose(m, pcm.astype(np.float64).mean(axis=1) / 32768.0, atol=1e-7)
    segs = io.ingest(pcm, 44100, 16000, 4096)
    assert segs.shape == (4, 4096) and segs.dtype == np.float32           # 16000 samples -> 4 segments of 4096
    assert np.all(segs.reshape(-1)[16000:] == 0) and np.any(segs.reshape(-1)[:16000] != 0)
    assert io.n_segments(0, 44100, 16000, 4096) == 1 and io.ingest(np.zeros((0, 1), np.float32), 44100, 16000, 4096).shape == (1, 4096)
    assert io.n_segments(32767 * 3, 16000, 16000, 32767) == 3 and io.n_segments(32767 * 3 + 1, 16000, 16000, 32767) == 4
    with pytest.raises(ValueError):
        io.mono(np.zeros((4, 1), np.float64))


def test_identity_rate_is_a_copy():
    x = np.random.default_rng(1).standard_normal(1000).astype(np.float32)
    assert np.array_equal(io.resample(x, 16000, 16000), x)
