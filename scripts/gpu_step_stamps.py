"""Where a decode step's time goes, from in-kernel wall-clock stamps (YMT3_STAMP=1): per kernel of one step at a late
position, the gap since the previous kernel's last exit, the dispatch ramp (first -> last workgroup entry), and the span
(first entry -> last exit).  BASELINE configs[1], 64 segments.  100 MHz clock: 0.01 us resolution."""
import os, sys
os.environ["YMT3_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yourmt3_amd.audio import synthetic_segments
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
cfg = baseline_config(1)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
m = YourMT3(cfg, max_batch=B)
a = torch.from_numpy(synthetic_segments(B, cfg.segment_samples)).cuda()
enc = m.encode(m.logmel(a))
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m.decode(enc, n_steps); torch.cuda.synchronize()
m.decode(enc, n_steps); torch.cuda.synchronize()
rows = m.step_stamps()
print(f"decode step at position {n_steps - 1} (graph replay), us from the first workgroup entry of the step")
print(f"{'kernel':16s} {'WGs':>5s} {'gap':>6s} {'ramp':>6s} {'span':>6s} {'first_exit':>10s}")
prev_out = None
tot_gap = tot_span = 0.0
for name, grid, in0, in1, out0, out1 in rows:
    gap = in0 - prev_out if prev_out is not None else 0.0
    print(f"{name:16s} {grid:5d} {gap:6.2f} {in1 - in0:6.2f} {out1 - in0:6.2f} {out0 - in0:10.2f}")
    tot_gap += gap; tot_span += out1 - in0
    prev_out = out1
print(f"step: {rows[-1][5] - rows[0][2]:.2f} us = spans {tot_span:.2f} + gaps {tot_gap:.2f}")

# distribution of workgroup exit times inside the two attention kernels of layer 3 
import numpy as np
names = [r[0] for r in rows]
nth = lambda name, n: [i for i, x in enumerate(names) if x == name][n]
for k in ((nth("self_attn", 3), nth("cross_attn", 3)) if B == 64 and "self_attn" in names else ()):   # (the attention pair has its own marks: gpu_pair_marks.py)
    name, grid = rows[k][0], rows[k][1]
    st = m.kernel_stamps(k, grid).astype(np.int64)
    st = st[st[:, 0] > 0]                                   # a launcher may use fewer workgroups than the slot reserves
    grid = st.shape[0]
    t0 = st[:, 0].min()
    ent, ex = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0
    dur = ex - ent
    print(f"{name}: exit percentiles us p0 {np.percentile(ex,0):.2f} p10 {np.percentile(ex,10):.2f} p50 {np.percentile(ex,50):.2f} p90 {np.percentile(ex,90):.2f} p100 {ex.max():.2f}; per-WG duration p50 {np.percentile(dur,50):.2f} p100 {dur.max():.2f}")
    idx = np.arange(grid)
    print("  mean exit by blockIdx % 8 :", " ".join(f"{ex[idx % 8 == x].mean():.2f}" for x in range(8)))
    print("  mean exit by head (idx % 8 is also the head); by row quartile:", " ".join(f"{ex[(idx // 8) // 16 == q].mean():.2f}" for q in range(4)))
    print("  mean entry by blockIdx % 8:", " ".join(f"{ent[idx % 8 == x].mean():.2f}" for x in range(8)))
    order = np.argsort(ex)
    print("  last 12 to exit (blockIdx: exit):", " ".join(f"{i}:{ex[i]:.2f}" for i in order[-12:]))
