"""Per-launch-class times of a decode step of BASELINE configs[i] (eager decode, HIP events every 16th step).  usage: gpu_r03_cfg_prof.py <i> <B> <L>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yourmt3_amd.audio import synthetic_segments
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
i, B, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = baseline_config(i)
m = YourMT3(cfg, max_batch=B)
a = torch.from_numpy(synthetic_segments(B, cfg.segment_samples)).cuda()
enc = m.encode(m.logmel(a))
prof = m.profile_decode(enc, L, stride=16)
n = L // 16
tot = 0.0
for k, v in prof.items():
    if v["launches"] and k != "unsampled_span":
        us = 1e3 * v["ms_total"] / v["launches"]
        per_step = v["launches"] / n
        tot += us * per_step
        print(f"{k:16s} {per_step:5.1f} launches/step  {us:7.2f} us each (with event overhead)  {us * per_step:8.1f} us/step")
sp = prof["unsampled_span"]
print("configs[%d] B=%d L=%d: bracketed sum %.1f us/step; true step (unbracketed) %.1f us" % (i, B, L, tot, 1e3 * sp["ms_total"] / (sp["launches"] * 15)))
