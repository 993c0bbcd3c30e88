"""Entry / exit stamps of every launch of one decode step (YMT3_STAMP=1) of BASELINE configs[i]: us from the step's first kernel.  usage: <i> <B> <L>"""
import os, sys
os.environ["YMT3_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yourmt3_amd.audio import synthetic_segments
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
i, B, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = baseline_config(i)
m = YourMT3(cfg, max_batch=B)
a = torch.from_numpy(synthetic_segments(B, cfg.segment_samples)).cuda()
enc = m.encode(m.logmel(a))
m.decode(enc, L); torch.cuda.synchronize()
rows = m.step_stamps()
prev_exit = None
for r in rows:
    gap = "" if prev_exit is None else "  gap %5.2f" % (r[2] - prev_exit)
    print("%-16s grid %5d  first entry %8.2f  last entry %8.2f  first exit %8.2f  last exit %8.2f  span %6.2f%s" % (r + (r[5] - r[2], gap)))
    prev_exit = r[5]
