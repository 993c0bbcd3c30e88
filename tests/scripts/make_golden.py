"""Generate tests/golden/*.npz from the CPU oracle (bf16-emulating mode = the HIP numerics contract).

The reference tree holds no fixtures (parity unpinned), so these vectors are produced by
this build's own oracle, which tests/test_oracle_vs_thirdparty.py pins against torch.stft and HF T5.
Run:  python tests/scripts/make_golden.py      (CPU only, ~1 min)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ymt3_oracle as O                     # noqa: E402
from yourmt3_amd.config import YMT3Config               # noqa: E402
from yourmt3_amd.weights import make_weights, f32_to_bf16_bits  # noqa: E402

CASES = {
    "small_t64": (YMT3Config(segment_samples=8191, max_decode_len=64), 2, 48),
    "full_t256": (YMT3Config(max_decode_len=128), 2, 128),
    "mc3_t64": (YMT3Config(segment_samples=8191, max_decode_len=32, n_channels=3), 2, 24),
}
LOGIT_STEPS = (0, 1, 23)
# The whole decode length (round 3): configs[1] shapes, 2 segments x 1024 steps, EOS fill off.  Encoder output = that of "full_t256" (same
# weights, same audio seed; the decode length does not enter the weights), so only ids, margins and logits at late positions are stored:
# 127 | 382, 383, 384 (the self-attention's first full 384-key block and its second loop iteration) | 767 | 1023.
LONG_CASE = ("full_t256_l1024", YMT3Config(max_decode_len=1024, eos_id=-1), 2, 1024)
LONG_LOGIT_STEPS = (0, 127, 382, 383, 384, 767, 1023)


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for name, (cfg, B, n_steps) in CASES.items():
        W = make_weights(cfg, seed=1234)
        audio = O.synthetic_audio(B, cfg, seed=0)
        mel, enc = O.encode(audio, W, cfg, bf16=True)
        toks, logits = O.greedy_decode(enc, W, cfg, n_steps, bf16=True, return_logits=True)
        top2 = logits.topk(2, -1).values
        margin = (top2[..., 0] - top2[..., 1]).numpy().astype(np.float32)
        np.savez_compressed(
            os.path.join(out_dir, name + ".npz"),
            mel=mel.numpy().astype(np.float32),
            enc_bf16=f32_to_bf16_bits(enc),
            tokens=toks.numpy().astype(np.int32),
            margin=margin,
            logit_steps=np.array(LOGIT_STEPS, dtype=np.int32),
            logits=logits[:, :, list(LOGIT_STEPS), :].numpy().astype(np.float32),
            n_steps=np.int32(n_steps), seed_weights=np.int32(1234), seed_audio=np.int32(0),
        )
        print(name, "tokens", toks.shape, "distinct", len(set(toks.flatten().tolist())), "min margin", float(margin.min()))
    name, cfg, B, n_steps = LONG_CASE
    W = make_weights(cfg, seed=1234)
    _, enc = O.encode(O.synthetic_audio(B, cfg, seed=0), W, cfg, bf16=True)
    toks, logits = O.greedy_decode(enc, W, cfg, n_steps, bf16=True, return_logits=True)
    top2 = logits.topk(2, -1).values
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), tokens=toks.numpy().astype(np.int32),
                        margin=(top2[..., 0] - top2[..., 1]).numpy().astype(np.float32),
                        logit_steps=np.array(LONG_LOGIT_STEPS, dtype=np.int32),
                        logits=logits[:, :, list(LONG_LOGIT_STEPS), :].numpy().astype(np.float32),
                        n_steps=np.int32(n_steps), seed_weights=np.int32(1234), seed_audio=np.int32(0), enc_from="full_t256")
    print(name, "tokens", toks.shape, "distinct", len(set(toks.flatten().tolist())))
    # integer relative-position bucket tables (a5), both directions
    q = np.arange(512)[:, None]
    k = np.arange(512)[None, :]
    np.savez_compressed(os.path.join(out_dir, "relpos_buckets.npz"),
                        bidirectional=O.relative_position_bucket(k - q, True, 32, 128).astype(np.int8),
                        causal=O.relative_position_bucket(k - q, False, 32, 128).astype(np.int8))


if __name__ == "__main__":
    main()
