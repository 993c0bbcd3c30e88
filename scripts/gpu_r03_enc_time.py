"""Encoder time (ymt3_encode, 64 segments, configs[1]) and per-GEMM times via rocprof-free HIP events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yourmt3_amd.audio import synthetic_segments
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
cfg = baseline_config(1)
m = YourMT3(cfg, max_batch=64)
a = torch.from_numpy(synthetic_segments(64, cfg.segment_samples)).cuda()
mel = m.logmel(a)
for _ in range(3): enc = m.encode(mel)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(20):
    e0.record(); enc = m.encode(mel); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ts.sort()
ms = ts[len(ts) // 2]
print("encode: median %.3f ms (min %.3f) -> %.0f TFLOP/s, encoder_whole %.3f; checksum %.6f" % (ms, ts[0], 10.47e9 * 64 / (ms * 1e-3) / 1e12, 10.47e9 * 64 / (ms * 1e-3) / 2.5e15, enc.float().abs().mean().item()))
