import sys, time; sys.path.insert(0, "/root/repo")
import torch
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
from yourmt3_amd.audio import synthetic_segments
cfg = baseline_config(3)
m = YourMT3(cfg, max_batch=64)
a = torch.from_numpy(synthetic_segments(64, cfg.segment_samples)).cuda()
m.inference(a, max_token_length=128); torch.cuda.synchronize()
t0 = time.perf_counter(); m.inference(a, max_token_length=128); torch.cuda.synchronize()
print("ms per 128 steps", 1e3 * (time.perf_counter() - t0))
