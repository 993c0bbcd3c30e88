#!/bin/bash
# Round 3: the MoE chain (moe_chain.hip): bit-identity + oracle parity, then configs[4] timing A/B (interleaved)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "moe" > gpurun_out/r03_moe_tests.log 2>&1; rc=$?
echo "pytest exit=$rc"; tail -12 gpurun_out/r03_moe_tests.log
[ $rc -ne 0 ] && exit $rc
cat > /tmp/moe_time.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from yourmt3_amd.audio import synthetic_segments
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
cfg = baseline_config(4)
if len(sys.argv) > 1 and sys.argv[1] == "bf16": cfg = cfg.with_(moe_fp8=0)
m = YourMT3(cfg, max_batch=64)
a = torch.from_numpy(synthetic_segments(64, cfg.segment_samples)).cuda()
m.inference(a); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2): m.inference(a)
torch.cuda.synchronize()
print("%.1f ms per batch, fallbacks %d" % (1e3 * (time.perf_counter() - t0) / 2, m.merged_fallbacks))
PY
for i in 1 2; do
  echo -n "fp8  five launches: "; YMT3_NO_MOE_CHAIN=1 timeout -k 10 120 python /tmp/moe_time.py || exit 1
  echo -n "fp8  MoE chain    : "; timeout -k 10 120 python /tmp/moe_time.py || exit 1
  echo -n "bf16 five launches: "; YMT3_NO_MOE_CHAIN=1 timeout -k 10 120 python /tmp/moe_time.py bf16 || exit 1
  echo -n "bf16 MoE chain    : "; timeout -k 10 120 python /tmp/moe_time.py bf16 || exit 1
done
