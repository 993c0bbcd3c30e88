"""Round 3: the merged decode kernels beyond 64 rows.  For each variant (environment at handle creation) time configs[1]'s shapes at several
batch sizes and digest the ids: every variant must produce the same ids (the merged kernels are bit-identical to the separate launches).
    python scripts/gpu_r03_rows.py [config_index=1] [B ...]"""
import hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yourmt3_amd.audio import synthetic_segments
from yourmt3_amd.config import baseline_config
from yourmt3_amd.model import YourMT3
ci = int(sys.argv[1]) if len(sys.argv) > 1 else 1
Bs = [int(x) for x in sys.argv[2:]] or [96, 128, 192, 256]
cfg = baseline_config(ci)
VARIANTS = {"merged kernels up to 64 rows (round 2)": {},
            "attention pair + GEMM chain up to 256 rows": {"YMT3_MERGED_MAX_ROWS": "256"},
            "the same, chain with 76 KB LDS (two workgroups per CU)": {"YMT3_MERGED_MAX_ROWS": "256", "YMT3_CHAIN_W2F": "1"},
            "attention pair up to 256 rows, GEMMs as launches": {"YMT3_MERGED_MAX_ROWS": "256", "YMT3_NO_GEMM_CHAIN": "1"}}
out = {}
for name, env in VARIANTS.items():
    if "YMT3_CHAIN_W2F" in env:
        continue           # (the switch is read once per process: run this variant as its own process with the variable exported)
    for k, v in env.items(): os.environ[k] = v
    m = YourMT3(cfg, max_batch=max(Bs))
    for k in env: del os.environ[k]
    rec = {}
    for B in Bs:
        a = torch.from_numpy(synthetic_segments(B, cfg.segment_samples)).cuda()
        t = m.inference(a); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2): t = m.inference(a)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        rec[f"B={B}"] = {"ms_per_batch": round(1e3 * dt, 1), "us_per_step": round(1e6 * dt / cfg.max_decode_len, 1), "audio_s_per_s": round(B * cfg.segment_seconds / dt, 1),
                         "ids_sha256": hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16], "fallbacks": m.merged_fallbacks}
        print(name, f"B={B}", rec[f"B={B}"], flush=True)
    out[name + (" [YMT3_CHAIN_W2F=1 exported]" if os.environ.get("YMT3_CHAIN_W2F") == "1" else "")] = rec
    m.close()
print(json.dumps(out, indent=1))
