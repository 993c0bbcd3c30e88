#!/bin/bash
# Round 3: in-situ counters for the encoder-side kernels of the bench workload (configs[1], 64 segments, T5 encoder) and of configs[2]
# (Perceiver-TF encoder, 256 segments): the torch-free host tools/ymt3_run runs the WHOLE path with a 4-step decode, so every encoder kernel is
# profiled with the shapes and the cache state it has in the benchmark.  One counter per pass, --kernel-trace only (MI355X_MICROARCH.md).
#   MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); HBM bytes = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB
export TMPDIR=/tmp
mkdir -p gpurun_out
RE="gemm_big_kernel|gemm_kernel|enc_attn_kernel|seq_attn_kernel|rmsnorm_kernel|spec_embed_kernel|logmel_kernel|f32_to_bf16|broadcast"
for c in 1 2; do
  python3 -m yourmt3_amd.export_blob /tmp/blob$c.bin $c > /dev/null || exit 1
  B=64; [ $c -eq 2 ] && B=256
  for ctr in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE; do
    d=gpurun_out/pmce_${c}_$ctr
    rm -rf $d
    timeout -k 10 200 rocprofv3 --pmc $ctr --kernel-trace --kernel-include-regex "$RE" --output-format csv -d $d -- tools/ymt3_run /tmp/blob$c.bin $B 4 2 0 $c > $d.log 2>&1
    rc=$?
    echo "config $c $ctr exit=$rc"
    [ $rc -ne 0 ] && { tail -5 $d.log; exit 1; }
    f=$(find $d -name "*counter_collection.csv" | head -1)
    python3 scripts/pmc_summary2.py "$f" $ctr > ${d}_summary.json || exit 1
    t=$(find $d -name "*kernel_trace.csv" | head -1)
    [ "$ctr" = GRBM_GUI_ACTIVE ] && [ -n "$t" ] && cp "$t" gpurun_out/pmce_${c}_kernel_trace.csv
    rm -rf $d
  done
  # per-kernel durations of the same run without counters
  d=gpurun_out/pmce_${c}_stats
  rm -rf $d
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- tools/ymt3_run /tmp/blob$c.bin $B 4 3 0 $c > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
  find $d -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} gpurun_out/r03_encoder_config${c}_kernel_stats.csv
  rm -rf $d
done
python3 scripts/pmc_encoder_merge.py
