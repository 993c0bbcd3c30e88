#!/bin/bash
# many-row GEMM experiments: parity of the many-row kernels, configs[3] launch classes, configs[3] / B=256 batch times
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "many_row or multichannel or channel" 2>&1 | tail -3 || exit 1
timeout -k 10 300 python scripts/gpu_r03_cfg_prof.py 3 64 256 2>&1 | grep -v amdgpu.ids
cat > /tmp/cfg_time.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from yourmt3_amd.audio import synthetic_segments
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
i, B, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = baseline_config(i)
m = YourMT3(cfg, max_batch=B)
a = torch.from_numpy(synthetic_segments(B, cfg.segment_samples)).cuda()
m.inference(a, max_token_length=L); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2): m.inference(a, max_token_length=L)
torch.cuda.synchronize()
print("configs[%d] B=%d L=%d: %.1f ms per batch" % (i, B, L, 1e3 * (time.perf_counter() - t0) / 2))
PY
timeout -k 10 200 python /tmp/cfg_time.py 3 64 256 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python /tmp/cfg_time.py 1 256 1024 2>&1 | grep -v amdgpu.ids
