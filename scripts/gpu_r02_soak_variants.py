"""Soak of the attention pair in its other regimes: MoE decoder (pair without the GEMM chain), 512-frame segments (two cross K/V blocks),
ragged batches; every repetition's ids must equal the first's, no abort poison."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yourmt3_amd.audio import synthetic_segments
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
bad = 0
for label, cfg, B, L, reps in (("configs[4] MoE fp8", baseline_config(4), 64, 1024, 8),
                               ("configs[1] 512 frames", baseline_config(1).with_(segment_samples=65535), 64, 1024, 8),
                               ("configs[1] B=37", baseline_config(1), 37, 512, 12)):
    m = YourMT3(cfg, max_batch=B)
    a = torch.from_numpy(synthetic_segments(B, cfg.segment_samples)).cuda()
    ref = m.inference(a, max_token_length=L)
    for i in range(reps):
        t = m.inference(a, max_token_length=L)
        if not torch.equal(t, ref) or int(t.min()) < 0:
            bad += 1
            print(label, "repetition", i, "differs", flush=True)
    print(label, "ok" if bad == 0 else "FAILED", flush=True)
    m.close()
sys.exit(1 if bad else 0)
