"""Stage marks inside one moe_chain launch (YMT3_STAMP=1): per workgroup, us from the kernel's first entry."""
import os, sys
os.environ["YMT3_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yourmt3_amd.audio import synthetic_segments
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
cfg = baseline_config(4)
if len(sys.argv) > 1 and sys.argv[1] == "bf16": cfg = cfg.with_(moe_fp8=0)
m = YourMT3(cfg, max_batch=64)
a = torch.from_numpy(synthetic_segments(64, cfg.segment_samples)).cuda()
enc = m.encode(m.logmel(a))
m.decode(enc, 512); torch.cuda.synchronize()
rows = m.step_stamps()
for r in rows: print("%-16s grid %4d  first entry %8.2f  last entry %8.2f  first exit %8.2f  last exit %8.2f" % r)
names = [r[0] for r in rows]
k = [i for i, n in enumerate(names) if n == "gemm_chain"][2]
raw = m.kernel_stamps(k, 256 + 1024).astype(np.int64).reshape(-1)
inout = raw[:512].reshape(256, 2)
marks = raw[512:512 + 2048].reshape(256, 8)
t0 = inout[:, 0].min()
us = lambda v: (v - t0) / 100.0
lab = ["s0 done+signalled", "router: tile arrived", "router done+signalled", "routing arrived (s2 start)", "s2 done+signalled", "expert's hidden arrived", "s3 done+signalled", "all y arrived (s4 start)"]
print("entry: min %.2f max %.2f; exit: min %.2f max %.2f" % (us(inout[:, 0]).min(), us(inout[:, 0]).max(), us(inout[:, 1]).min(), us(inout[:, 1]).max()))
for j, l in enumerate(lab):
    v = marks[:, j]; v = v[v > 0]
    print(f"{l:28s} n={len(v):3d} min {us(v).min():6.2f} median {np.median(us(v)):6.2f} max {us(v).max():6.2f}")
