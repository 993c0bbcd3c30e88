"""Experiment: two handles of 32 segments each, decoding concurrently with the merged kernels (each handle believes it is alone), against one handle
of 64.  usage: gpu_r03_two_handles.py [L]   (YMT3_CHAIN_W2F=1 recommended: 76 KB chain workgroups share a CU with the attention pair's)"""
import os, sys, time, threading, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yourmt3_amd.audio import synthetic_segments
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = baseline_config(1)
a = torch.from_numpy(synthetic_segments(64, cfg.segment_samples)).cuda()
one = YourMT3(cfg, max_batch=64)
enc = one.encode(one.logmel(a))
ref = one.decode(enc, L); torch.cuda.synchronize()
t0 = time.perf_counter(); ref = one.decode(enc, L); torch.cuda.synchronize()
t_one = time.perf_counter() - t0
m = [YourMT3(cfg, max_batch=32), YourMT3(cfg, max_batch=32)]
halves = [enc[:32].contiguous(), enc[32:].contiguous()]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
out = [None, None]
def run(i):
    with torch.cuda.stream(streams[i]):
        out[i] = m[i].decode(halves[i], L)
def both():
    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
both()
t0 = time.perf_counter(); both(); t_two = time.perf_counter() - t0
got = torch.cat([out[0], out[1]])
print("one handle of 64: %.1f ms; two handles of 32 concurrently: %.1f ms; ids equal: %s; fallbacks %d %d" %
      (1e3 * t_one, 1e3 * t_two, bool(torch.equal(got.cpu(), ref.cpu())), m[0].merged_fallbacks, m[1].merged_fallbacks))
t0 = time.perf_counter(); run(0); torch.cuda.synchronize(); print("one handle of 32 alone: %.1f ms" % (1e3 * (time.perf_counter() - t0)))
