"""Stage marks inside one attn_pair launch (YMT3_STAMP=1): per workgroup, us from the kernel's first entry."""
import os, sys
os.environ["YMT3_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yourmt3_amd.audio import synthetic_segments
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
cfg = baseline_config(1)
m = YourMT3(cfg, max_batch=64)
a = torch.from_numpy(synthetic_segments(64, cfg.segment_samples)).cuda()
enc = m.encode(m.logmel(a))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m.decode(enc, n); torch.cuda.synchronize()
rows = m.step_stamps()
names = [r[0] for r in rows]
k = [i for i, nm in enumerate(names) if nm == "attn_pair"][2]
raw = m.kernel_stamps(k, 512 + 2048).astype(np.int64).reshape(-1)
inout = raw[:1024].reshape(512, 2)
marks = raw[1024:1024 + 4096].reshape(512, 8)
t0 = marks[:, 0].min()
us = lambda v: (v - t0) / 100.0
lab = ["entry", "self stream consumed", "partial stored", "row signalled", "row complete", "cross block consumed"]
print(f"position {n - 1}; first-loads-issued stamp: min %.2f max %.2f; exit: min %.2f max %.2f" % (us(inout[:, 0]).min(), us(inout[:, 0]).max(), us(inout[:, 1]).min(), us(inout[:, 1]).max()))
fl = np.sort(us(inout[:, 0]) - us(marks[:, 0]))
print("first-loads-issued stamp - entry (the stamp waits for the scalar loads in front of it: kernarg words and the step counter): p10 %.2f median %.2f p90 %.2f max %.2f" % (fl[51], fl[256], fl[460], fl[-1]))
for j, l in enumerate(lab):
    v = marks[:, j]
    print(f"{l:22s} min {us(v).min():6.2f} p10 {np.percentile(us(v), 10):6.2f} median {np.median(us(v)):6.2f} p90 {np.percentile(us(v), 90):6.2f} max {us(v).max():6.2f}")
life = (inout[:, 1] - marks[:, 0]) / 100.0
print("workgroup lifetime: min %.2f median %.2f max %.2f" % (life.min(), np.median(life), life.max()))
for j in range(1, 6):
    d = (marks[:, j] - marks[:, j - 1]) / 100.0
    print(f"  {lab[j - 1]} -> {lab[j]}: median {np.median(d):.2f} p90 {np.percentile(d, 90):.2f}")
d = (inout[:, 1] - marks[:, 5]) / 100.0
print(f"  cross block consumed -> exit: median {np.median(d):.2f} p90 {np.percentile(d, 90):.2f}")
order = np.argsort(marks[:, 0])
print("entry order vs blockIdx (first 16):", order[:16].tolist())
