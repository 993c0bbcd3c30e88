#!/bin/bash
# Perceiver-TF encoder: two library builds, encoder output digest and time at configs[2] (B = 256)
set -o pipefail
cat > /tmp/ptf_time.py <<'PY'
import os, sys, time, hashlib, torch
sys.path.insert(0, os.getcwd())
from yourmt3_amd.audio import synthetic_segments
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
cfg = baseline_config(2)
m = YourMT3(cfg, max_batch=256)
a = torch.from_numpy(synthetic_segments(256, cfg.segment_samples)).cuda()
mel = m.logmel(a)
enc = m.encode(mel); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    e0.record(); enc = m.encode(mel); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print("encoder %.2f ms (min of 5)  digest %s" % (min(ts), hashlib.sha1(enc.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]))
PY
for v in before after before after; do echo -n "$v: "; YMT3_LIB=$PWD/gpurun_ab/lib_$v.so timeout -k 10 300 python /tmp/ptf_time.py 2>&1 | grep -v amdgpu.ids || exit 1; done
