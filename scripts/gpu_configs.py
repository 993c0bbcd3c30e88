"""Throughput of the other BASELINE.json configs on one GPU (secondary numbers; the headline is bench.py)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
from yourmt3_amd.audio import synthetic_segments

def throughput(cfg, B, L, reps=2):
    m = YourMT3(cfg, max_batch=B)
    a = torch.from_numpy(synthetic_segments(B, cfg.segment_samples)).cuda()
    m.inference(a, max_token_length=L); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): m.inference(a, max_token_length=L)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    gb = m.device_bytes / 1e9
    m.close()
    return {"ms_per_batch": round(1e3 * dt, 1), "audio_s_per_s": round(B * cfg.segment_seconds / dt, 1), "device_GB": round(gb, 2)}

out = {
    "configs[1] T5 enc, B=64, L=1024": throughput(baseline_config(1), 64, 1024),
    "configs[2] Perceiver-TF enc (32 latents per frame, 3 blocks), B=256, L=1024": throughput(baseline_config(2), 256, 1024),
    "configs[3] 13-channel decoder, B=64, L=256": throughput(baseline_config(3), 64, 256),
    "configs[4] MoE decoder (8 experts, top-2, fp8 expert GEMMs), B=64, L=1024": throughput(baseline_config(4), 64, 1024),
    "configs[1] at B=256": throughput(baseline_config(1), 256, 1024),
    "configs[1] with 512-frame segments (4.096 s), B=64, L=1024": throughput(baseline_config(1).with_(segment_samples=65535), 64, 1024),
}
print(json.dumps(out, indent=1))
