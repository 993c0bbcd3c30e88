"""Secondary shapes on the GPU box: 512-frame segments (S=65535) and the 13-channel decoder (config 4 shape).
Parity (teacher-forced, short) + throughput.  Not the headline metric."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yourmt3_amd.config import YMT3Config, baseline_config
from yourmt3_amd.model import YourMT3
from yourmt3_amd.weights import make_weights
from oracle import ymt3_oracle as O

def parity(cfg, B, n):
    W = make_weights(cfg)
    m = YourMT3(cfg, W, max_batch=B)
    a = O.synthetic_audio(B, cfg)
    mel_o, enc_o = O.encode(a, W, cfg, True)
    mel = m.logmel(a.cuda()); enc = m.encode(mel)
    t_o, l_o = O.greedy_decode(enc_o, W, cfg, n, True, return_logits=True)
    t_g, l_g = m.decode(enc_o.bfloat16().cuda(), n, forced=t_o.cuda(), return_logits=True)
    top2 = l_o.topk(2, -1).values; safe = (top2[..., 0] - top2[..., 1]) >= 0.06
    res = {"mel": (mel.cpu() - mel_o).abs().max().item(), "enc": (enc.float().cpu() - enc_o).abs().max().item(),
           "logits": (l_g.cpu() - l_o).abs().max().item(), "argmax_ok": bool(torch.equal(t_g.cpu()[safe], t_o[safe])), "safe_frac": safe.float().mean().item()}
    m.close()
    return res

def throughput(cfg, B, L, reps=2):
    m = YourMT3(cfg, max_batch=B)
    a = O.synthetic_audio(B, cfg).cuda()
    m.inference(a, max_token_length=L); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): m.inference(a, max_token_length=L)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    m.close()
    return {"ms_per_batch": 1e3 * dt, "audio_s_per_s": B * cfg.segment_seconds / dt, "device_GB": None}

out = {}
c512 = YMT3Config(segment_samples=65535, max_decode_len=1024, eos_id=-1)
out["t512_parity"] = parity(c512.with_(max_decode_len=32), 2, 24)
out["t512_b64_l1024"] = throughput(c512, 64, 1024)
c13 = baseline_config(3)
out["mc13_parity"] = parity(c13.with_(max_decode_len=16), 6, 12)      # 78 rows -> MT=1 path
out["mc13_parity_rows_gt_128"] = parity(c13.with_(max_decode_len=16, segment_samples=8191), 12, 8)   # 156 rows -> MT=4 path
out["mc13_b64_l256"] = throughput(c13, 64, 256)
print(json.dumps(out, indent=1))
