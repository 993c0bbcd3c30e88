"""Soak test of the merged decode kernels (attention pair, GEMM chain): many batches back to back, every batch's ids must equal the
first batch's (same audio) and none may carry the abort poison; then the same through the slot scheduler."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yourmt3_amd.audio import synthetic_segments
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cfg = baseline_config(1)
m = YourMT3(cfg, max_batch=64)
a = torch.from_numpy(synthetic_segments(64, cfg.segment_samples)).cuda()
ref = m.inference(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
bad = 0
for i in range(n):
    t = m.inference(a)
    if not torch.equal(t, ref) or int(t.min()) < 0:
        bad += 1
        print("batch", i, "differs: min id", int(t.min()), "differing ids", int((t != ref).sum()), flush=True)
    if i % 20 == 19:
        torch.cuda.synchronize()
        print(f"{i + 1} batches, {1e3 * (time.perf_counter() - t0) / (i + 1):.1f} ms per batch, {bad} bad", flush=True)
s = m.inference_stream(a, slots=48, interval=16)
print("slot scheduler (48 slots) equals lock-step:", bool(torch.equal(s, ref)))
print("launched decode steps per step kernels: ok" if bad == 0 else "FAILED")
sys.exit(1 if bad else 0)
