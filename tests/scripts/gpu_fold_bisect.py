"""Where does the folded O-projection first differ from the separate launch?  (debug; GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import ymt3_oracle as O
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
from yourmt3_amd.weights import make_weights

cfg = baseline_config(1)
W = make_weights(cfg, seed=1234)
for B, L in ((4, 128), (64, 64), (16, 64), (32, 64), (4, 700)):
    os.environ["YMT3_NO_FOLD_O"] = "1"
    old = YourMT3(cfg, W, device=0, max_batch=B)
    del os.environ["YMT3_NO_FOLD_O"]
    new = YourMT3(cfg, W, device=0, max_batch=B)
    a = O.synthetic_audio(B, cfg, seed=0).cuda()
    e = old.encode(old.logmel(a))
    t_old, l_old = old.decode(e, L, return_logits=True)
    t_new, l_new = new.decode(e, L, forced=t_old, return_logits=True)
    d = (l_new - l_old).abs().amax(-1)[:, 0]            # (B, L)
    bad = (d > 0).nonzero()
    print(f"B={B} L={L}: max diff {float(d.max()):.3e}, differing (row, step) pairs {bad.shape[0]}",
          "first by step:", sorted(bad.tolist(), key=lambda x: (x[1], x[0]))[:6], flush=True)
    if bad.shape[0]:
        rows = sorted(set(int(x) for x in bad[:, 0].tolist()))
        print("   rows affected:", rows[:40], "steps min", int(bad[:, 1].min()))
    old.close(); new.close()
