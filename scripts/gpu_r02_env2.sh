#!/bin/bash
# round 2: graph replay vs eager launches x where the kernel arguments live (the dispatch now fetches 14 dwords of them per launch)
python3 -m yourmt3_amd.export_blob /tmp/blob.bin 1 || exit 1
run() { echo "== $*"; env "$@" timeout -k 5 90 tools/ymt3_run /tmp/blob.bin 64 1024 3 | grep "pass [12]"; }
run A=1
run HIP_FORCE_DEV_KERNARG=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 HIP_FORCE_DEV_KERNARG=1
run YMT3_NO_GRAPH=1
run YMT3_NO_GRAPH=1 HIP_FORCE_DEV_KERNARG=1
run YMT3_NO_GRAPH=1 HIP_FORCE_DEV_KERNARG=1 GPU_MAX_HW_QUEUES=1
run YMT3_NO_GRAPH=1 HIP_FORCE_DEV_KERNARG=1 AMD_DIRECT_DISPATCH=0
for e in "A=1" "YMT3_NO_GRAPH=1 HIP_FORCE_DEV_KERNARG=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 HIP_FORCE_DEV_KERNARG=1"; do
  echo "== bench.py $e"; env $e timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('rtf', round(d['value'],1), 'ms', round(d['ms_per_step'],1))"
done
