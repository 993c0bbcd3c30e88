#!/bin/bash
# at 256 rows the per-(row, head) weight pulls of the fused cross-attention (wq) and the folded self-attention (wo) scale with
# the rows: 268 MB of L2 -> CU traffic per layer.  Are the separate GEMM launches cheaper there?
cat > /tmp/bv.py <<'PY'
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
print({k: os.environ.get(k) for k in ("YMT3_NO_FOLD_O", "YMT3_NO_FUSEQ")}, " B=256 L=512:", thr(baseline_config(1), 256, 512), "ms;  B=128 L=512:", thr(baseline_config(1), 128, 512), "ms", flush=True)
PY
for e in "A=1" "YMT3_NO_FOLD_O=1" "YMT3_NO_FUSEQ=1"; do env $e timeout -k 10 300 python /tmp/bv.py 2>&1 | grep -v amdgpu.ids; done
