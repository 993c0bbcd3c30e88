#!/bin/bash
# round 2: where do kernel arguments live (now that the dispatch fetches them)?  then the PMC traffic passes of the attention kernels
python3 -m yourmt3_amd.export_blob /tmp/blob.bin 1 || exit 1
run() { echo "== $*"; env "$@" timeout -k 5 60 tools/ymt3_run /tmp/blob.bin 64 512 3 | grep "pass [12]"; }
run A=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run YMT3_NO_GRAPH=1
run YMT3_NO_GRAPH=1 HIP_FORCE_DEV_KERNARG=1
bash scripts/gpu_pmc.sh && python3 scripts/pmc_collect.py
timeout -k 10 600 python bench.py > gpurun_out/r02_bench_line.json 2> gpurun_out/bench.err; echo "bench exit=$?"
python -c "import json; d=json.loads(open('gpurun_out/r02_bench_line.json').read().strip().splitlines()[-1]); print({k: d[k] for k in ('value','ms_per_step','p50_segment_latency_ms')}, d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
