#!/bin/bash
# does the 4-deep mid-size tile kernel pay at 256 rows (B = 256 single channel)?
cat > /tmp/b256.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
from yourmt3_amd.audio import synthetic_segments
def thr(cfg, B, L):
    m = YourMT3(cfg, max_batch=B)
    a = torch.from_numpy(synthetic_segments(B, cfg.segment_samples)).cuda()
    m.inference(a, max_token_length=L); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2): m.inference(a, max_token_length=L)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    m.close()
    return round(1e3 * dt, 1)
print("mid rows", os.environ.get("YMT3_DEC_GEMM_MID_ROWS"), " B=256 L=512:", thr(baseline_config(1), 256, 512), "ms;  B=128 L=512:", thr(baseline_config(1), 128, 512), "ms", flush=True)
PY
for t in 0 256 128; do YMT3_DEC_GEMM_MID_ROWS=$t timeout -k 10 300 python /tmp/b256.py 2>&1 | grep -v amdgpu.ids; done
