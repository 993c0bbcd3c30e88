"""Latency of small batches (BASELINE configs[0]'s shape on the GPU path: one 2.048 s segment, 1024 tokens) and the
batch-size curve at the configs[1] shapes.  Output: JSON (profiles/r01_small_batches.json, r02_small_batches.json)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yourmt3_amd.audio import synthetic_segments
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
cfg = baseline_config(1)
m = YourMT3(cfg, max_batch=128)
out = {}
for B in (1, 2, 4, 8, 16, 32, 64, 128):
    a = torch.from_numpy(synthetic_segments(B, cfg.segment_samples)).cuda()
    m.inference(a); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2): m.inference(a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    out[f"B={B}"] = {"ms_per_batch": round(1e3 * dt, 1), "us_per_step": round(1e6 * dt / cfg.max_decode_len, 1),
                     "audio_s_per_s": round(B * cfg.segment_seconds / dt, 1), "ms_per_segment": round(1e3 * dt / B, 2)}
print(json.dumps(out, indent=1))
m.close()
