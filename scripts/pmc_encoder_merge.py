"""Merge the per-counter summaries of scripts/gpu_r03_pmc_encoder.sh into gpurun_out/r03_pmc_encoder.json (copied to profiles/)."""
import csv, json, os, re
d = "gpurun_out/"
out = {"source": "scripts/gpu_r03_pmc_encoder.sh: rocprofv3 7.2 --pmc <one counter> --kernel-trace (separate passes) around `tools/ymt3_run blob B 4 2 0 config`: "
                 "the whole hot path with a 4-step decode, two passes per process, so every encoder-side kernel runs with the bench's shapes",
       "definitions": {"mfma_utilisation": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): busy cycles are summed over all SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs",
                       "hbm_bytes": "2 x FETCH_SIZE (gfx950 correction for wide coalesced streaming reads, MI355X_MICROARCH.md) + WRITE_SIZE, KiB -> bytes",
                       "flops": "2 M N K of the GEMM the grid size identifies; attention 4 T^2 d per (sequence, head)"}}
for c, label in ((1, "configs[1]: T5 encoder, 64 segments x 256 frames (M = 16384 rows)"), (2, "configs[2]: Perceiver-TF encoder, 256 segments (8.4 M spectral tokens, 2.1 M latent rows)")):
    try:
        S = {k: json.load(open(f"{d}pmce_{c}_{k}_summary.json"))["kernels"] for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE")}
    except FileNotFoundError:
        continue
    kern = {}
    tot_busy = tot_act = 0.0
    for name, m in S["SQ_VALU_MFMA_BUSY_CYCLES"].items():
        g = S["GRBM_GUI_ACTIVE"].get(name)
        if not g:
            continue
        short = re.sub(r"void \(anonymous namespace\)::", "", name)
        act = g["avg_per_launch"] / 8.0
        rec = {"launches": m["launches"], "mfma_busy_cycles_per_launch": m["avg_per_launch"], "gpu_active_cycles_per_launch_per_xcd": act,
               "mfma_utilisation": round(m["avg_per_launch"] / (1024.0 * act), 4) if act else None}
        f, w = S["FETCH_SIZE"].get(name), S["WRITE_SIZE"].get(name)
        if f and w:
            rec["hbm_bytes_per_launch"] = 2.0 * f["avg_per_launch"] * 1024 + w["avg_per_launch"] * 1024
        kern[short] = rec
        tot_busy += m["sum"]; tot_act += g["sum"] / 8.0
    out[label] = {"kernels": kern, "all_encoder_side_kernels": {"mfma_busy_cycles": tot_busy, "gpu_active_cycles_per_xcd": tot_act,
                                                                 "mfma_utilisation": round(tot_busy / (1024.0 * tot_act), 4) if tot_act else None}}
    st = f"{d}r03_encoder_config{c}_kernel_stats.csv"
    if os.path.exists(st):
        rows = list(csv.DictReader(open(st)))
        out[label]["kernel_stats"] = [{"name": re.sub(r"void \(anonymous namespace\)::", "", r["Name"]), "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2),
                                       "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 3), "pct": float(r["Percentage"])} for r in rows[:14]]
json.dump(out, open(d + "r03_pmc_encoder.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
