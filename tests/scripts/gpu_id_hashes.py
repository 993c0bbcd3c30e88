"""Free-running token-id digests of the HIP path on seeded inputs (GPU box).  Run once on the round-1 kernels to write
tests/golden/r01_id_hashes.json; tests/test_gpu_parity.py::test_ids_bit_identical_to_round_1 recomputes them, so any change to
the summation order of a decode kernel shows up as a changed digest (free-running greedy ids over hundreds of steps move
with the last bit of a logit).

    python tests/scripts/gpu_id_hashes.py [--write] [case names ...]
"""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from oracle import ymt3_oracle as O  # noqa: E402
from yourmt3_amd.config import YMT3Config, baseline_config, FFN_MOE, ENC_PERCEIVER_TF  # noqa: E402
from yourmt3_amd.model import YourMT3  # noqa: E402
from yourmt3_amd.weights import make_weights  # noqa: E402

PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "r01_id_hashes.json")

# name -> (config, segments, decode steps, audio seed)
CASES = {
    "configs1_full_b64_l1024_seed21": (baseline_config(1), 64, 1024, 21),
    "configs1_full_b64_l1024_seed0": (baseline_config(1), 64, 1024, 0),
    "small_t64_b4_l64": (YMT3Config(segment_samples=8191, max_decode_len=64, eos_id=-1), 4, 64, 7),
    "mc3_t64_b4_l32": (YMT3Config(segment_samples=8191, max_decode_len=32, n_channels=3, eos_id=-1), 4, 32, 7),
    "mc13_t128_b3_l16": (YMT3Config(segment_samples=16383, max_decode_len=16, n_channels=13, eos_id=-1), 3, 16, 0),
    "t512_b2_l64": (YMT3Config(segment_samples=65535, max_decode_len=64, eos_id=-1), 2, 64, 0),
    "moe_bf16_t64_b3_l32": (YMT3Config(segment_samples=8191, max_decode_len=32, dec_ffn=FFN_MOE, eos_id=-1), 3, 32, 0),
    "moe_fp8_t64_b3_l32": (YMT3Config(segment_samples=8191, max_decode_len=32, dec_ffn=FFN_MOE, moe_fp8=1, eos_id=-1), 3, 32, 0),
}


# the MoE cases pin the round-1 sums, which include the combine launch's own sum(h^2) tree: the round-2 default folds the combine
# into the next norm GEMM (another tree, same tolerance against the oracle), so these cases run with the launch kept
ENV = {"moe_bf16_t64_b3_l32": {"YMT3_MOE_COMBINE_LAUNCH": "1"}, "moe_fp8_t64_b3_l32": {"YMT3_MOE_COMBINE_LAUNCH": "1"}}


def digest(cfg, B, L, seed, env=None):
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        m = YourMT3(cfg, make_weights(cfg, seed=1234), device=0, max_batch=B)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    a = O.synthetic_audio(B, cfg, seed=seed).cuda()
    t = m.inference(a, max_token_length=L).cpu().contiguous()
    m.close()
    return hashlib.sha256(t.numpy().tobytes()).hexdigest()


def main():
    got = {}
    only = [a for a in sys.argv[1:] if not a.startswith("--")]           # optional: case names to run
    for name, (cfg, B, L, seed) in CASES.items():
        if only and name not in only:
            continue
        got[name] = digest(cfg, B, L, seed, ENV.get(name))
        print(name, got[name], flush=True)
    if "--write" in sys.argv:
        os.makedirs("gpurun_out", exist_ok=True)
        with open(os.path.join("gpurun_out", "r01_id_hashes.json"), "w") as f:
            json.dump(got, f, indent=1)
    elif os.path.exists(PATH):
        ref = json.load(open(PATH))
        bad = [k for k in got if ref.get(k) != got[k]]
        print("MISMATCH: " + ", ".join(bad) if bad else "all digests equal the round-1 ones")
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
