#!/bin/bash
# HBM traffic counters for the round-2 decode kernels (attention pair, GEMM chain): one counter per pass, kernel-trace only, the C host,
# four 256-position pieces -- exactly as scripts/gpu_pmc.sh (see there and profiles/r02_notes.md for why pieces).
export TMPDIR=/tmp
export YMT3_DEBUG_HOOKS=1
mkdir -p gpurun_out
python3 -m yourmt3_amd.export_blob /tmp/blob.bin 1 || exit 1
for ctr in FETCH_SIZE WRITE_SIZE; do
  for half in 0 256 512 768; do
    d=gpurun_out/pmcp_${ctr}_$half
    rm -rf $d
    timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --kernel-include-regex "dec_attn_pair_kernel|dec_chain_kernel" --output-format csv -d $d -- tools/ymt3_run /tmp/blob.bin 64 256 1 $half > $d.log 2>&1
    rc=$?
    echo "$ctr half=$half exit=$rc"
    [ $rc -ne 0 ] && { tail -5 $d.log; exit 1; }
    f=$(find $d -name "*counter_collection.csv" | head -1)
    python3 scripts/pmc_summary.py "$f" $ctr > ${d}_summary.json || exit 1
    rm -rf $d
  done
done
python3 - <<'PY'
import json
d = "gpurun_out/"
pieces = [0, 256, 512, 768]
load = lambda n: json.load(open(d + n))["kernels"]
F = {p: load(f"pmcp_FETCH_SIZE_{p}_summary.json") for p in pieces}
W = {p: load(f"pmcp_WRITE_SIZE_{p}_summary.json") for p in pieces}
names = list(F[0].keys())
out = {"source": "rocprofv3 7.2 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only, --kernel-include-regex 'dec_attn_pair_kernel|dec_chain_kernel') "
                 "around `tools/ymt3_run blob 64 256 1 <step0>` for step0 = 0, 256, 512, 768 (scripts/gpu_pmc_pair.sh)",
       "units": "FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 correction for wide coalesced streaming reads)",
       "config": "BASELINE configs[1], 64 segments, 1024 positions, 6 decoder layers"}
def agg(match):
    ks = [n for n in names if match in n]
    fs = sum(F[p][k]["sum"] for p in pieces for k in ks); n = sum(F[p][k]["launches"] for p in pieces for k in ks)
    ws = sum(W[p][k]["sum"] for p in pieces for k in ks); wn = sum(W[p][k]["launches"] for p in pieces for k in ks)
    fb, wb = 2.0 * fs * 1024 / n, ws * 1024 / wn
    return {"launches": n, "fetch_bytes_per_launch_corrected": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
out["attn_pair"] = agg("dec_attn_pair_kernel")
out["attn_pair"]["algorithmic_bytes_per_launch"] = 64 * 8 * 512.5 * 64 * 2 * 2 + 64 * 8 * 256 * 64 * 2 * 2
out["attn_pair"]["note"] = ("algorithmic = self K/V cache (67.2 MB averaged over positions) + cross K/V (33.55 MB); on top of that the kernel reads its head's 64 KB of wo and "
                            "64 KB of wq per workgroup (L2-served after first touch) and hands 1 MB of O-projection partial rows from head to head at agent scope "
                            "(written through, read eight times)")
out["gemm_chain"] = agg("dec_chain_kernel")
out["gemm_chain"]["algorithmic_bytes_per_launch"] = 2 * (512 * 512 + 2048 * 512 + 512 * 2048 + 1536 * 512)
out["gemm_chain"]["note"] = "algorithmic = the four weight matrices (6.0 MB); activations, sum(h^2) partials and the FFN hidden cross the stages at agent scope"
json.dump(out, open("gpurun_out/r02_pmc_attn_pair.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
