#!/bin/bash
# counter-line stride experiment: MoE chain A/B + the headline bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "moe_chain or bit_identical or 1024_positions" 2>&1 | tail -3 || exit 1
timeout -k 10 200 python scripts/gpu_moe_chain_marks.py 2>&1 | grep -v amdgpu.ids | tail -10
cat > /tmp/moe_time.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from yourmt3_amd.audio import synthetic_segments
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
cfg = baseline_config(4)
if sys.argv[1] == "bf16": cfg = cfg.with_(moe_fp8=0)
m = YourMT3(cfg, max_batch=64)
a = torch.from_numpy(synthetic_segments(64, cfg.segment_samples)).cuda()
m.inference(a); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2): m.inference(a)
torch.cuda.synchronize()
print("%s no_chain=%s: %.1f ms per batch, fallbacks %d" % (sys.argv[1], os.environ.get("YMT3_NO_MOE_CHAIN", "0"), 1e3 * (time.perf_counter() - t0) / 2, m.merged_fallbacks))
PY
for i in 1 2; do for k in fp8; do for v in 0; do YMT3_NO_MOE_CHAIN=$v timeout -k 10 120 python /tmp/moe_time.py $k 2>&1 | grep -v amdgpu.ids || exit 1; done; done; done
timeout -k 10 300 python bench.py --steps 3 --warmup 1 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench ms_per_step', d['ms_per_step'], 'value', d['value'])"
