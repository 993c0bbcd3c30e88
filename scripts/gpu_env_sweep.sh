#!/bin/bash
# HIP runtime knobs vs the decode loop's launch gaps: the torch-free C host, 64 segments, 512 steps, 3 passes each.
python3 -m yourmt3_amd.export_blob /tmp/blob.bin 1 || exit 1
run() { echo "== $*"; env "$@" timeout -k 5 60 tools/ymt3_run /tmp/blob.bin 64 512 3 | grep "pass [12]"; }
run A=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=3
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run GPU_MAX_HW_QUEUES=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=64
run AMD_DIRECT_DISPATCH=0
run YMT3_NO_GRAPH=1
