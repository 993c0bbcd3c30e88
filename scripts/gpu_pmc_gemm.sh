#!/bin/bash
# MFMA-busy and GPU-active cycle counters for the encoder GEMM kernels (separate --pmc passes, kernel-trace only, on the
# torch-free tools/gemm_probe).  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles).
export TMPDIR=/tmp
mkdir -p gpurun_out
for ctr in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES; do
  d=gpurun_out/pmc_gemm_$ctr
  rm -rf $d
  timeout -k 10 120 rocprofv3 --pmc $ctr --kernel-trace --kernel-include-regex "gemm_big_kernel" --output-format csv -d $d -- tools/gemm_probe > $d.log 2>&1
  echo "$ctr exit=$?"
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 scripts/pmc_summary.py "$f" $ctr > ${d}_summary.json && cat ${d}_summary.json
  t=$(find $d -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && cp "$t" gpurun_out/pmc_gemm_${ctr}_trace.csv
  rm -rf $d
done
