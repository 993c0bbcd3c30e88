#!/bin/bash
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chains" 2>&1 | tail -2 || exit 1
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
t = m.inference(a, max_token_length=L); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2): m.inference(a, max_token_length=L)
torch.cuda.synchronize()
import hashlib
print("configs[%d] B=%d L=%d chains=%s: %.1f ms per batch  ids %s" % (i, B, L, os.environ.get("YMT3_CHAINS", "1"), 1e3 * (time.perf_counter() - t0) / 2, hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:12]))
PY
for B in 96 112 128; do for c in 1 2; do YMT3_CHAINS=$c timeout -k 10 300 python /tmp/cfg_time.py 1 $B 512 2>&1 | grep -v amdgpu.ids || exit 1; done; done
for B in 176 256; do timeout -k 10 300 python /tmp/cfg_time.py 1 $B 512 2>&1 | grep -v amdgpu.ids | sed 's/$/ (default)/' || exit 1; done
