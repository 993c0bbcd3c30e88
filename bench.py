#!/usr/bin/env python3
"""Headline benchmark: audio-seconds transcribed per wall-second (RTF) on BASELINE.json configs[1]
(MT3 base / T5-small, 2.048 s 128-mel segments, batch 64 per GPU, 1024-token greedy decode).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" is one pass of the whole hot path (audio already resident in HBM -> log-mel -> encoder ->
cross-KV -> 1024 greedy decode steps -> token ids, then the all-gather of the token streams across
ranks).  Weak scaling: every rank transcribes its own 64 segments, no data-path collective except
that all-gather.  Rank 0 prints ONE JSON line (contract: task prompt / DESIGN.md "Measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from yourmt3_amd.audio import synthetic_segments  # noqa: E402
from yourmt3_amd.config import baseline_config  # noqa: E402
from yourmt3_amd.dist import init_distributed, shard_range, all_gather_tokens  # noqa: E402
from yourmt3_amd.model import YourMT3  # noqa: E402

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def synthetic_audio(n: int, cfg, seed: int, device) -> torch.Tensor:
    """(n, S) fp32 synthetic segments (yourmt3_amd.audio.synthetic_segments)."""
    return torch.from_numpy(synthetic_segments(n, cfg.segment_samples, cfg.sample_rate, seed)).to(device)


def self_attn_algorithmic_bytes(cfg, rows: int, t: int) -> int:
    """K and V of every (row, head), t+1 cached keys x 64 x bf16, one decoder layer."""
    return rows * cfg.n_heads * (t + 1) * cfg.d_kv * 2 * 2


def pmc_traffic(cfg, B: int, L: int):
    """HBM bytes per self-attention launch from the committed rocprofv3 PMC passes (profiles/, produced by
    scripts/gpu_pmc.sh; FETCH_SIZE doubled per the gfx950 correction).  None if the profile does not match."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_decode_attn.json")
    if not os.path.exists(path) or B != 64 or L != 1024 or cfg.n_channels != 1:
        return None
    with open(path) as f:
        return json.load(f)["self_attn"]["hbm_bytes_per_launch"]


def cpu_baseline(cfg, sample_segments: int, sample_steps: int):
    """The oracle (a CPU *port* of this path, there being no reference implementation) on host cores."""
    from oracle import ymt3_oracle as O
    from yourmt3_amd.weights import make_weights
    W = make_weights(cfg, seed=1234)
    a = O.synthetic_audio(sample_segments, cfg, seed=0)
    # tiny per-step ops: more threads than ~16 only add fork/join overhead (128 threads ran 15x slower)
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    O.transcribe_segments(a, W, cfg, n_steps=sample_steps, bf16=False)
    dt = time.perf_counter() - t0
    # scale the audio credited by the fraction of the decode that was run (decode dominates; stated in `sample`)
    audio_s = sample_segments * cfg.segment_seconds * (sample_steps / cfg.max_decode_len)
    return {
        "value": audio_s / dt, "unit": "audio_s/wall_s", "cores": cores, "kind": "port",
        "sample": f"{sample_segments} segment(s), first {sample_steps} of {cfg.max_decode_len} decode steps, fp32 oracle, "
                  f"{dt:.1f} s wall; audio credited pro rata to decode steps",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="segments per GPU (BASELINE configs[1]: 64)")
    ap.add_argument("--decode-len", type=int, default=0, help="override L (0 = config's 1024); non-default runs are not the headline metric")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--profile-stride", type=int, default=32)
    args = ap.parse_args()

    rank, world, local_rank = init_distributed(args.gpus)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    cfg = baseline_config(1)
    L = args.decode_len or cfg.max_decode_len
    B = args.batch
    model = YourMT3(cfg, seed=1234, device=local_rank, max_batch=B)
    lo, hi = shard_range(B * world, rank, world)            # contiguous block split of the global segment batch
    audio = synthetic_audio(B * world, cfg, seed=0, device="cpu")[lo:hi].to(dev)
    assert audio.shape[0] == B

    def step():
        toks = model.inference(audio, max_token_length=L)
        return all_gather_tokens(toks, world)

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    lat = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        s0 = time.perf_counter()
        out = step()
        torch.cuda.synchronize(dev)
        lat.append(time.perf_counter() - s0)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    assert out.shape == (B * world, cfg.n_channels, L)

    audio_seconds = args.steps * world * B * cfg.segment_seconds
    result = {
        "metric": "audio-sec transcribed / wall-sec (RTF)",
        "value": audio_seconds / elapsed,
        "unit": "audio_s/wall_s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "p50_segment_latency_ms": 1e3 * statistics.median(lat) / B,
        "p50_batch_latency_ms": 1e3 * statistics.median(lat),
        "config": {
            "workload": "BASELINE configs[1]: MT3 base (T5-small dims, 6+6 layers, d512, 8x64 heads, d_ff 2048, vocab 1536), "
                        "2.048 s / 16 kHz / 128-mel segments (256 encoder frames), greedy decode forced to "
                        f"{L} tokens, seeded random weights",
            "segments_per_gpu": B, "global_batch": B * world, "decode_len": L, "frames": cfg.n_frames,
            "parallelism": f"dp{world} (segments sharded, token ids all-gathered over RCCL)",
        },
    }

    if rank == 0 and not args.no_roofline:
        # dominant kernel: decoder self-attention (streams the growing KV cache).  Timed live with HIP
        # events on the launch stream, every `stride`-th step of one eager decode of the same batch.
        mel = model.logmel(audio)
        enc = model.encode(mel)
        prof = model.profile_decode(enc, L, stride=args.profile_stride)
        sa = prof["self_attn"]
        sampled_t = [t for t in range(L) if t % args.profile_stride == args.profile_stride // 2]
        rows = B * cfg.n_channels
        bytes_total = sum(self_attn_algorithmic_bytes(cfg, rows, t) for t in sampled_t) * cfg.n_dec_layers
        assert sa["launches"] == len(sampled_t) * cfg.n_dec_layers, (sa, len(sampled_t))
        # An event pair costs stream time of its own, so per-launch brackets over-read.  Calibrate against the true step
        # time (one bracket around the stride-1 un-bracketed steps after every sampled step): the per-bracket overhead
        # is (sum of all brackets of a sampled step - true step time) / launches per step.
        span = prof["unsampled_span"]
        kern = {k: v for k, v in prof.items() if k != "unsampled_span"}
        n_sampled = len(sampled_t)
        step_true_ms = span["ms_total"] / max(1, span["launches"] * (args.profile_stride - 1))
        launches_per_step = sum(v["launches"] for v in kern.values()) / n_sampled
        pair_ms = max(0.0, (sum(v["ms_total"] for v in kern.values()) / n_sampled - step_true_ms) / launches_per_step)
        avg_ms = sa["ms_total"] / sa["launches"] - pair_ms
        achieved = (bytes_total / sa["launches"]) / (avg_ms * 1e-3) / 1e9
        result["roofline"] = {
            "kernel": "dec_attn_kernel<true> (decoder self-attention over the KV cache)",
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc_traffic(cfg, B, L),
            "avg_launch_us": 1e3 * avg_ms, "launches_timed": sa["launches"], "event_pair_overhead_us": 1e3 * pair_ms,
            "eager_step_us": 1e3 * step_true_ms,
            "algorithmic_bytes_per_launch": bytes_total / sa["launches"],
            "note": "bytes = rows*heads*(t+1)*64*2B*2 (K and V) averaged over sampled positions t = stride/2, 3*stride/2, ...; "
                    "duration = HIP events around each sampled launch on the launch stream minus the per-bracket overhead calibrated against un-bracketed steps; traffic = FETCH_SIZE*2 + "
                    "WRITE_SIZE per launch from profiles/r01_pmc_decode_attn.json (separate rocprofv3 --pmc passes)",
        }
        step_ms = {k: max(0.0, v["ms_total"] / max(1, v["launches"]) - pair_ms) * (cfg.n_dec_layers if k not in ("lm_head_gemm", "argmax_embed") else 1)
                   for k, v in kern.items()}
        result["decode_step_breakdown_us"] = {k: round(1e3 * v, 2) for k, v in step_ms.items()}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(baseline_config(0).with_(eos_id=-1), sample_segments=16, sample_steps=1024)

    if rank == 0:
        print(json.dumps(result), flush=True)
    model.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
