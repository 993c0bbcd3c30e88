"""Merge the per-piece summaries scripts/gpu_pmc.sh wrote under gpurun_out/ into profiles/r02_pmc_decode_attn.json."""
import json, os, sys
d = "gpurun_out/"
pieces = [0, 256, 512, 768]
def load(n): return json.load(open(d + n))["kernels"]
F = {p: load(f"pmc_FETCH_SIZE_{p}_summary.json") for p in pieces}
W = {p: load(f"pmc_WRITE_SIZE_{p}_summary.json") for p in pieces}
names = list(F[0].keys())
SELF = [n for n in names if "<true" in n][0]
CROSS = [n for n in names if "<false" in n][0]
out = {"source": "rocprofv3 7.2 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only, --kernel-include-regex dec_attn_kernel) "
                 "around `tools/ymt3_run blob 64 256 1 <step0>` for step0 = 0, 256, 512, 768 (scripts/gpu_pmc.sh; the debug hook "
                 "ymt3_debug_decode_start lets a short process cover late positions: the profiler's counter mode aborts on longer runs, profiles/r02_notes.md)",
       "units": "FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced streaming reads; "
                "re-checked here with tools/pmc_probe.cpp: 524298 KiB reported for a 1048576 KiB float4 copy read)",
       "config": "BASELINE configs[1], 64 segments, 1024 positions, 6 decoder layers; round-2 kernels: the self-attention kernel also reads its head's 64 KB of "
                 "wo (O-projection folded in) and writes a 2 KB partial row, the cross-attention kernel includes the fused query projection and reads eight partial rows",
       "pieces": {}}
for p in pieces:
    out["pieces"][f"t{p}_{p + 255}"] = {"FETCH_SIZE": {k.split("::")[1]: v for k, v in F[p].items()},
                                          "WRITE_SIZE": {k.split("::")[1]: v for k, v in W[p].items()}}
for label, k in (("self_attn", SELF), ("cross_attn", CROSS)):
    fs = sum(F[p][k]["sum"] for p in pieces); n = sum(F[p][k]["launches"] for p in pieces)
    ws = sum(W[p][k]["sum"] for p in pieces); wn = sum(W[p][k]["launches"] for p in pieces)
    fb, wb = 2.0 * fs * 1024 / n, ws * 1024 / wn
    out[label] = {"launches": n, "fetch_bytes_per_launch_corrected": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
out["self_attn"]["algorithmic_bytes_per_launch"] = 64 * 8 * 512.5 * 64 * 2 * 2
out["self_attn"]["note"] = "algorithmic = the K/V cache alone (67.2 MB averaged over positions); the folded O-projection adds 33.55 MB of wo reads per launch that are L2-served after first touch and 1 MB of partial rows written"
out["cross_attn"]["algorithmic_bytes_per_launch"] = 64 * 8 * 256 * 64 * 2 * 2 + 64 * 8 * 64 * 512 * 2
out["cross_attn"]["note"] = "algorithmic = K/V slab (33.55 MB) + the head's 64 x 512 bf16 query-projection weights per workgroup (33.55 MB, L2-served after first touch)"
json.dump(out, open("profiles/r02_pmc_decode_attn.json", "w"), indent=1)
json.dump(out, open("gpurun_out/r02_pmc_decode_attn.json", "w"), indent=1)      # gpurun merges gpurun_out/ back; profiles/ on the box is not
print(json.dumps({k: out[k] for k in ("self_attn", "cross_attn")}, indent=1))
