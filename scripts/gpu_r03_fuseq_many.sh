#!/bin/bash
set -o pipefail
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
for B in 128 256; do
  timeout -k 10 300 python /tmp/cfg_time.py 1 $B 512 2>&1 | grep -v amdgpu.ids || exit 1
  YMT3_NO_FUSEQ=1 timeout -k 10 300 python /tmp/cfg_time.py 1 $B 512 2>&1 | grep -v amdgpu.ids | sed 's/$/  (separate query projection)/' || exit 1
done
