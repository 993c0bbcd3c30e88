#!/bin/bash
# do the kernels of two concurrent decode chains overlap in time?  kernel trace of a short run, overlap computed from the timestamps
export TMPDIR=/tmp
python3 -m yourmt3_amd.export_blob /tmp/blob.bin 1 || exit 1
rm -rf gpurun_out/prof_chain
YMT3_CHAINS=2 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_chain -- tools/ymt3_run /tmp/blob.bin 64 96 1 > gpurun_out/prof_chain.log 2>&1; echo "exit=$?"
f=$(find gpurun_out/prof_chain -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "dec_" in r["Kernel_Name"] or "argmax" in r["Kernel_Name"]]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"][:40]) for r in rows)
print("decode kernels:", len(ev), "queues:", collections.Counter(e[2] for e in ev))
# time with >= 2 kernels in flight / time with >= 1
pts = []
for s, e, q, n in ev: pts += [(s, 1), (e, -1)]
pts.sort()
busy1 = busy2 = 0; depth = 0; last = pts[0][0]
for t, d in pts:
    if depth >= 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    depth += d; last = t
print("time with >=1 kernel running: %.1f ms, with >=2: %.1f ms (%.1f %%)" % (busy1 / 1e6, busy2 / 1e6, 100.0 * busy2 / max(1, busy1)))
span = ev[-1][1] - ev[0][0]
print("span of the decode: %.1f ms; sum of kernel durations: %.1f ms" % (span / 1e6, sum(e - s for s, e, _, _ in ev) / 1e6))
for s, e, q, n in ev[2000:2024]: print(q, n, (s - ev[2000][0]) / 1e3, (e - s) / 1e3)
PY
rm -rf gpurun_out/prof_chain
