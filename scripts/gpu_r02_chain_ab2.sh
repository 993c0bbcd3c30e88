#!/bin/bash
# GEMM chain A/B: with and without the stamp buffer, and decode-only times at two lengths
set -o pipefail
for st in 0 1; do
for v in 0 1; do
  YMT3_STAMP=$st YMT3_NO_GEMM_CHAIN=$v timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('stamp=$st no_gemm_chain=$v', 'rtf', round(d['value'],1), 'ms', round(d['ms_per_step'],2))" || { tail -5 gpurun_out/bench_err.log; exit 1; }
done
done
cat > /tmp/dec_ab.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
from yourmt3_amd.audio import synthetic_segments
cfg = baseline_config(1)
m = YourMT3(cfg, max_batch=64)
a = torch.from_numpy(synthetic_segments(64, cfg.segment_samples)).cuda()
enc = m.encode(m.logmel(a))
for n in (128, 1024):
    m.decode(enc, n); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): m.decode(enc, n)
    torch.cuda.synchronize()
    print(f"decode {n} steps: {1e6 * (time.perf_counter() - t0) / 3 / n:.1f} us per step", flush=True)
PY
for v in 0 1 0 1; do YMT3_NO_GEMM_CHAIN=$v timeout -k 10 200 python /tmp/dec_ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/no_gemm_chain=$v /"; done
