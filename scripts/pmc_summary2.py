"""Reduce a rocprofv3 counter_collection.csv to sums per (kernel, grid size) -- one counter per pass.  The encoder's GEMM shapes share template
instances (gemm_big_kernel<EPI>), so the grid size is what tells the QKV projection from the FFN-in projection."""
import csv, json, sys, collections
path, ctr = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); cnt = collections.Counter()
with open(path, newline="") as f:
    for row in csv.DictReader(f):
        if row.get("Counter_Name") != ctr:
            continue
        k = row["Kernel_Name"] + " | grid " + row.get("Grid_Size", "?") + " wg " + row.get("Workgroup_Size", "?")
        tot[k] += float(row["Counter_Value"]); cnt[k] += 1
out = {k: {"launches": cnt[k], "sum": tot[k], "avg_per_launch": tot[k] / cnt[k]} for k in tot}
print(json.dumps({"counter": ctr, "kernels": out}, indent=1))
