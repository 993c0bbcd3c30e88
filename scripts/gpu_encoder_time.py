"""Encoder-side kernel timing (HIP events around ymt3_encode) for A/B of the dense GEMM variants."""
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
for _ in range(3): m.encode(mel)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): enc = m.encode(mel)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"encoder {ms*1e3:.1f} us per batch -> {670.0/ms/1e3*1e3:.0f} TFLOP/s on the 670 GFLOP of GEMM+attention work; checksum {enc.float().abs().mean().item():.6f}")
