"""Per-launch-class times of a configs[4] decode step (eager decode, HIP events every 16th step): MoE chain vs the five launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yourmt3_amd.audio import synthetic_segments
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
cfg = baseline_config(4)
if len(sys.argv) > 1 and sys.argv[1] == "bf16": cfg = cfg.with_(moe_fp8=0)
m = YourMT3(cfg, max_batch=64)
a = torch.from_numpy(synthetic_segments(64, cfg.segment_samples)).cuda()
enc = m.encode(m.logmel(a))
prof = m.profile_decode(enc, 512, stride=16)
n = 512 // 16
tot = 0.0
for k, v in prof.items():
    if v["launches"] and k != "unsampled_span":
        us = 1e3 * v["ms_total"] / v["launches"]
        per_step = v["launches"] / n
        tot += us * per_step
        print(f"{k:16s} {per_step:5.1f} launches/step  {us:7.2f} us each (with event overhead)  {us * per_step:8.1f} us/step")
sp = prof["unsampled_span"]
print("bracketed sum %.1f us/step; true step (unbracketed) %.1f us" % (tot, 1e3 * sp["ms_total"] / (sp["launches"] * 15)))
