#!/bin/bash
# HBM traffic counters for the decode attention kernels: one counter per pass, kernel-trace only (as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes).  Uses the torch-free C host (tools/ymt3_run), restricts
# collection to the attention kernels and covers the 1024 positions as four 256-step pieces
# (debug hook ymt3_debug_decode_start, accepted under YMT3_DEBUG_HOOKS=1): rocprofv3 7.2 --pmc aborts on longer runs, see
# profiles/r02_notes.md for the attribution.
export TMPDIR=/tmp
export YMT3_DEBUG_HOOKS=1
mkdir -p gpurun_out
python3 -m yourmt3_amd.export_blob /tmp/blob.bin 1 || exit 1
for ctr in FETCH_SIZE WRITE_SIZE; do
  for half in 0 256 512 768; do
    d=gpurun_out/pmc_${ctr}_$half
    rm -rf $d
    timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --kernel-include-regex "dec_attn_kernel" --output-format csv -d $d -- tools/ymt3_run /tmp/blob.bin 64 256 1 $half > $d.log 2>&1
    echo "$ctr half=$half exit=$?"
    f=$(find $d -name "*counter_collection.csv" | head -1)
    python3 scripts/pmc_summary.py "$f" $ctr > ${d}_summary.json && cat ${d}_summary.json
    rm -rf $d
  done
done
