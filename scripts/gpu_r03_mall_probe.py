"""Is the attention pair faster when its K/V is served by the 256 MB memory-side cache?  A 1-layer decoder re-reads the same 100 MB every step (fits);
the 6-layer one cycles through 600 MB (does not).  Per-launch-class times, eager decode with HIP events every 16th step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yourmt3_amd.audio import synthetic_segments
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
for nl in (6, 1, 2):
    cfg = baseline_config(1).with_(n_dec_layers=nl)
    m = YourMT3(cfg, max_batch=64)
    a = torch.from_numpy(synthetic_segments(64, cfg.segment_samples)).cuda()
    enc = m.encode(m.logmel(a))
    prof = m.profile_decode(enc, 1024, stride=16)
    line = []
    for k, v in prof.items():
        if v["launches"] and k != "unsampled_span":
            line.append("%s %.2f us" % (k, 1e3 * v["ms_total"] / v["launches"]))
    sp = prof["unsampled_span"]
    print("decoder layers %d: %s; true step %.1f us" % (nl, "; ".join(line), 1e3 * sp["ms_total"] / (sp["launches"] * 15)))
    m.close()
