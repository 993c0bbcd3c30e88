"""Marks inside one dec_step_kernel launch (YMT3_STAMP=1): per layer, us from the kernel's first entry, over the workgroups of each row tile."""
import os, sys
os.environ["YMT3_STAMP"] = "1"
os.environ["YMT3_STEP_KERNEL"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yourmt3_amd.audio import synthetic_segments
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
cfg = baseline_config(1)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
m = YourMT3(cfg, max_batch=B)
a = torch.from_numpy(synthetic_segments(B, cfg.segment_samples)).cuda()
enc = m.encode(m.logmel(a))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m.decode(enc, n); torch.cuda.synchronize()
rows = m.step_stamps()
for r in rows:
    print("%-16s grid %4d  first entry %8.2f  last entry %8.2f  first exit %8.2f  last exit %8.2f" % r)
k = [i for i, r in enumerate(rows) if r[0] == "step_layers"][0]
G = rows[k][1]
raw = m.kernel_stamps(k, 8192).astype(np.int64).reshape(-1)
inout = raw[:2 * G].reshape(G, 2)
marks = raw[1024:1024 + 24 * G].reshape(G, 6, 4)
t0 = inout[:, 0].min()
us = lambda v: (v - t0) / 100.0
names = ["q there", "attention left", "tile's attention arrived", "chain tile done"]
chain = (np.arange(G) % 128) < 64
print(f"position {n - 1}, {B} rows; kernel span {us(inout[:, 1].max()):.2f} us")
for l in range(6):
    for j, nm in enumerate(names):
        sel = np.ones(G, bool) if j < 2 else chain
        v = marks[sel, l, j]
        v = v[v > 0]
        if v.size == 0:
            continue
        per_tile = " | ".join("%6.2f..%6.2f" % (us(marks[(np.arange(G) // 128 == t) & sel, l, j].min()), us(marks[(np.arange(G) // 128 == t) & sel, l, j].max())) for t in range(G // 128))
        print(f"layer {l} {nm:26s} min {us(v.min()):7.2f} median {np.median(us(v)):7.2f} max {us(v.max()):7.2f}   per row tile: {per_tile}")
