#!/usr/bin/env python3
"""Headline benchmark: audio-seconds transcribed per wall-second (RTF) on BASELINE.json configs[1]
(MT3 base / T5-small, 2.048 s 128-mel segments, batch 64 per GPU, 1024-token greedy decode).

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts N ranks itself, one per GPU)
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

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_BF16_PEAK_TFLOPS = 2500.0   # same guide: dense bf16 MFMA peak


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="segments per GPU (BASELINE configs[1]: 64)")
    ap.add_argument("--decode-len", type=int, default=0, help="override L (0 = config's 1024); non-default runs are not the headline metric")
    ap.add_argument("--frames", type=int, default=256, choices=(256, 512),
                    help="encoder frames per segment: 256 = the 2.048 s headline shape; 512 = BASELINE configs[1]'s '512-frame encoder' wording (4.096 s segments)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--profile-stride", type=int, default=32)
    return ap.parse_args(argv)


def synthetic_audio(n: int, cfg, seed: int, device):
    """(n, S) fp32 synthetic segments (yourmt3_amd.audio.synthetic_segments)."""
    import torch
    from yourmt3_amd.audio import synthetic_segments
    return torch.from_numpy(synthetic_segments(n, cfg.segment_samples, cfg.sample_rate, seed)).to(device)


# ----------------------------------------------------------------------------------------------------------------------
# algorithmic bytes, all bf16
def self_attn_bytes(cfg, rows: int, t: int) -> int:
    """K and V of every (row, head), t+1 cached keys x 64 x bf16, one decoder layer, position t."""
    return rows * cfg.n_heads * (t + 1) * cfg.d_kv * 2 * 2


def cross_attn_bytes(cfg, segments: int) -> int:
    """K and V slabs of every (segment, head), n_frames keys, one decoder layer."""
    return segments * cfg.n_heads * cfg.n_frames * cfg.d_kv * 2 * 2


def decoder_weight_bytes(cfg) -> dict:
    d, inner, ff = cfg.d_model, cfg.inner, cfg.d_ff
    per_layer = {"qkv_cache_gemm": 3 * inner * d, "self_o_gemm": d * inner, "cross_q": inner * d, "cross_o_gemm": d * inner,
                 "ffn_wi_gemm": ff * d, "ffn_wo_gemm": d * ff}
    out = {k: 2 * v for k, v in per_layer.items()}
    out["lm_head_gemm"] = 2 * cfg.vocab * d
    return out


def decode_bytes_per_batch(cfg, segments: int, L: int) -> int:
    """What one batch's decode must move at least once per step: every decoder weight, the cross K/V, the self K/V cache."""
    w = decoder_weight_bytes(cfg)
    per_step_w = sum(v for k, v in w.items() if k != "lm_head_gemm") * cfg.n_dec_layers + w["lm_head_gemm"]
    rows = segments * cfg.n_channels
    self_total = sum(self_attn_bytes(cfg, rows, t) for t in range(L)) * cfg.n_dec_layers
    return L * (per_step_w + cross_attn_bytes(cfg, segments) * cfg.n_dec_layers) + self_total


def encoder_flops_per_segment(cfg) -> float:
    """per token per layer 4 d^2 * 2 (QKVO) + 2 d d_ff * 2 (FFN) + 4 T d (scores + AV)."""
    T, d = cfg.n_frames, cfg.d_model
    return float(cfg.n_enc_layers * T * (8 * d * d + 4 * d * cfg.d_ff + 4 * T * d))


def pmc_file(name: str):
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


def pmc_traffic(cfg, B: int, L: int, paired: bool = False):
    """HBM bytes per launch of the dominant attention kernel from the committed rocprofv3 PMC passes (profiles/, produced by
    scripts/gpu_pmc.sh; FETCH_SIZE doubled per the gfx950 correction).  None if no profile matches the shape / kernel."""
    if B != 64 or L != 1024 or cfg.n_channels != 1 or cfg.n_frames != 256:
        return None
    if paired:
        d = pmc_file("r03_pmc_attn_pair.json") or pmc_file("r02_pmc_attn_pair.json")      # (the kernel is round 2's, unchanged)
        return None if d is None else d["attn_pair"]["hbm_bytes_per_launch"]
    d = pmc_file("r02_pmc_decode_attn.json") or pmc_file("r01_pmc_decode_attn.json")
    return None if d is None else d["self_attn"]["hbm_bytes_per_launch"]


# ----------------------------------------------------------------------------------------------------------------------
def cpu_model_name() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline():
    """The oracle (a CPU *port* of this path, there being no reference implementation) on the host cores: BASELINE
    configs[0] (one segment) and a 16-segment batch, each on ONE thread and on all useful threads.
    The headline case (16 segments, all useful threads) is timed in full; the others run the front-end + encoder fully and two
    bounded windows of the decode (first and last positions): mean per-step cost x 1024 (the cost grows linearly with the position)."""
    import torch
    from oracle import ymt3_oracle as O
    from yourmt3_amd.config import baseline_config
    from yourmt3_amd.weights import make_weights
    cfg = baseline_config(0).with_(eos_id=-1)
    W = make_weights(cfg, seed=1234)
    n_cpu = os.cpu_count() or 1
    # tiny per-step ops: more threads than ~16 only add fork/join overhead (128 threads ran 15x slower in round 1)
    many = min(16, n_cpu)
    L = cfg.max_decode_len
    # (label, segments, threads, decode steps timed per window; L = the whole decode).  A decode step costs more the later it comes
    # (the self-attention reads t cached keys and the cache append copies them), linearly in t: the bounded cases time a window at
    # the first positions and one at the last positions (cache pre-filled to L - n keys) and take the mean per-step cost times L;
    # the headline case (16 segments on all useful threads) runs all L steps
    cases = [("configs[0]: 1 segment", 1, 1, 48), ("configs[0]: 1 segment", 1, many, 48),
             ("16 segments", 16, 1, 10), ("16 segments", 16, many, L)]
    out = []
    for label, B, threads, n in cases:
        torch.set_num_threads(threads)
        a = O.synthetic_audio(B, cfg, seed=0)
        t0 = time.perf_counter()
        _, enc = O.encode(a, W, cfg, False)
        ckv = O.cross_kv(enc, W, cfg, False)
        t_enc = time.perf_counter() - t0

        def window(pos0):
            st = O.DecoderState(B, cfg)
            if pos0:
                g = torch.Generator().manual_seed(pos0)
                st.k = [0.1 * torch.randn(B, cfg.n_heads, pos0, cfg.d_kv, generator=g) for _ in range(cfg.n_dec_layers)]
                st.v = [0.1 * torch.randn(B, cfg.n_heads, pos0, cfg.d_kv, generator=g) for _ in range(cfg.n_dec_layers)]
                st.pos = pos0
            cur = torch.full((B,), cfg.pad_id, dtype=torch.long)
            w0 = time.perf_counter()
            for _ in range(n):
                cur = torch.argmax(O.decoder_step(cur, st, ckv, W, cfg, False), dim=-1)
            return (time.perf_counter() - w0) / n

        if n >= L:                                    # the headline case: the whole decode, nothing extrapolated
            w0 = time.perf_counter()
            O.greedy_decode(enc, W, cfg, L, False)
            t_dec = time.perf_counter() - w0
            rec = {"method": f"all {L} decode steps timed", "decode_steps_timed": L, "decode_s": round(t_dec, 2)}
        else:
            c_first, c_last = window(0), window(L - n)
            t_dec = L * 0.5 * (c_first + c_last)
            rec = {"method": f"{n} steps at the first and {n} at the last positions, mean per-step cost x {L}", "decode_steps_timed": 2 * n,
                   "ms_per_decode_step_first": round(1e3 * c_first, 3), "ms_per_decode_step_last": round(1e3 * c_last, 3),
                   "decode_s": round(t_dec, 2)}
        est = t_enc + t_dec
        out.append({"workload": label, "segments": B, "threads": threads, "value": B * cfg.segment_seconds / est, "encode_s": round(t_enc, 3), **rec})
    torch.set_num_threads(many)
    head = out[-1]
    return {
        "value": head["value"], "unit": "audio_s/wall_s", "cores": many, "kind": "port",
        "cpu_model": cpu_model_name(), "host_logical_cpus": n_cpu,
        "sample": f"fp32 oracle (oracle/ymt3_oracle.py), 16 segments on {many} threads, the whole path (front-end, encoder, all {L} decode "
                  f"steps) timed; `variants` adds configs[0] (1 segment) on 1 and {many} threads and the 16-segment batch on 1 thread, each with "
                  "the decode extrapolated from a window at the first and one at the last positions",
        "variants": out,
    }


# ----------------------------------------------------------------------------------------------------------------------
def run(args):
    import torch
    from yourmt3_amd.config import baseline_config
    from yourmt3_amd.dist import init_distributed, shard_range, all_gather_tokens, gather_floats
    from yourmt3_amd.model import YourMT3

    rank, world, local_rank = init_distributed(args.gpus)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    cfg = baseline_config(1)
    if args.frames == 512:
        cfg = cfg.with_(segment_samples=65535)
    L = args.decode_len or cfg.max_decode_len
    B = args.batch
    model = YourMT3(cfg, seed=1234, device=local_rank, max_batch=B)
    lo, hi = shard_range(B * world, rank, world)            # contiguous block split of the global segment batch
    audio = synthetic_audio(B * world, cfg, seed=0, device="cpu")[lo:hi].to(dev)
    assert audio.shape[0] == B

    def step():
        """one pass of the hot path + the all-gather of the ids; returns (ids, seconds of this rank's own work, seconds in the all-gather)"""
        s0 = time.perf_counter()
        toks = model.inference(audio, max_token_length=L)
        torch.cuda.synchronize(dev)          # (the decode call has waited for its own work already: include/ymt3.h, ymt3_set_abort_recovery)
        s1 = time.perf_counter()
        gathered = all_gather_tokens(toks, world)
        torch.cuda.synchronize(dev)
        return gathered, s1 - s0, time.perf_counter() - s1

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    lat, own, coll = [], [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, t_own, t_coll = step()
        lat.append(t_own + t_coll)
        own.append(t_own)
        coll.append(t_coll)
    t_rank = time.perf_counter() - t0            # this rank's K steps, before it waits for the others
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    # every rank's own times travel to rank 0's line: ms per step, ms of its own work (front-end .. ids), ms inside the all-gather
    # (= waiting for the slowest rank + the transfer itself)
    per_rank = gather_floats([1e3 * t_rank / args.steps, 1e3 * statistics.mean(own), 1e3 * statistics.mean(coll),
                              1e3 * min(coll), float(model.merged_fallbacks)], world, dev)
    assert out.shape == (B * world, cfg.n_channels, L)

    audio_seconds = args.steps * world * B * cfg.segment_seconds
    sec_per_batch = elapsed / args.steps
    result = {
        "metric": "audio-sec transcribed / wall-sec (RTF)",
        "value": audio_seconds / elapsed,
        "unit": "audio_s/wall_s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * sec_per_batch,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "p50_segment_latency_ms": 1e3 * statistics.median(lat) / B,
        "p50_batch_latency_ms": 1e3 * statistics.median(lat),
        "per_rank": {
            "ms_per_step": [round(r[0], 3) for r in per_rank],
            "ms_own_work": [round(r[1], 3) for r in per_rank],
            "ms_all_gather_mean": [round(r[2], 3) for r in per_rank],
            "ms_all_gather_min": [round(r[3], 3) for r in per_rank],
            "merged_kernel_fallbacks": [int(r[4]) for r in per_rank],
            "note": "one entry per rank: its own K steps / K (before the closing barrier); its own work per step (audio in HBM -> ids, "
                    "synchronised); mean and minimum time per step inside the RCCL all-gather of the ids (waiting for the slowest rank + "
                    "the transfer: the minimum over steps is close to the transfer alone); whether the rank left the merged decode kernels",
        },
        "config": {
            "workload": "BASELINE configs[1]: MT3 base (T5-small dims, 6+6 layers, d512, 8x64 heads, d_ff 2048, vocab 1536), "
                        f"{cfg.segment_seconds:.3f} s / 16 kHz / 128-mel segments ({cfg.n_frames} encoder frames), greedy decode forced to "
                        f"{L} tokens, seeded random weights",
            "segments_per_gpu": B, "global_batch": B * world, "decode_len": L, "frames": cfg.n_frames,
            "parallelism": f"dp{world} (segments sharded, token ids all-gathered over RCCL)",
        },
    }

    if rank == 0 and not args.no_roofline:
        result.update(roofline_block(model, cfg, audio, B, L, args.profile_stride, sec_per_batch))

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline()

    if rank == 0:
        print(json.dumps(result), flush=True)
    model.close()
    if world > 1:
        torch.distributed.barrier()             # rank 0 profiles after the timed region: leave the group together
        torch.distributed.destroy_process_group()


def roofline_block(model, cfg, audio, B, L, stride, sec_per_batch) -> dict:
    """Live per-kernel-class timing of one eager decode of the same batch (HIP events on the launch stream, every
    `stride`-th position, calibrated against un-bracketed steps), priced against the gfx950 peaks."""
    import torch
    out = {}
    mel = model.logmel(audio)
    # encoder: MFMA-bound.  Timed live with events around ymt3_encode on the launch stream.
    enc = model.encode(mel)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        enc = model.encode(mel)
    e1.record()
    torch.cuda.synchronize()
    enc_ms = e0.elapsed_time(e1) / reps
    enc_tflops = encoder_flops_per_segment(cfg) * B / (enc_ms * 1e-3) / 1e12
    gm = pmc_file("r03_pmc_encoder.json")
    gemm_pmc = None
    if gm is not None:
        sect = next((v for k, v in gm.items() if k.startswith("configs[1]")), None)
        if sect is not None:
            label = {"gemm_big_kernel<1>": "qkv_gemm", "gemm_big_kernel<2>": "ffn_in_gemm", "gemm_big_kernel<3>": "o_and_ffn_out_gemms", "gemm_big_kernel<4>": "cross_kv_gemm",
                     "gemm_big_kernel<0>": "mel_projection_gemm", "enc_attn_kernel<256>": "encoder_attention"}
            gemm_pmc = {lab: rec["mfma_utilisation"] for name, rec in sect["kernels"].items() for key, lab in label.items() if name.startswith(key)}
            gemm_pmc["all_encoder_side_kernels"] = sect["all_encoder_side_kernels"]["mfma_utilisation"]
    out["mfma_util"] = {
        "encoder_whole": enc_tflops / MFMA_BF16_PEAK_TFLOPS, "encoder_tflops": enc_tflops, "encoder_ms": enc_ms,
        "peak_tflops": MFMA_BF16_PEAK_TFLOPS,
        "gemm_pmc": gemm_pmc,
        "note": "encoder_whole = 10.47 GFLOP x segments / live time of ymt3_encode (front-end excluded; norms, attention and "
                "epilogues included) / 2.5 PFLOP/s dense bf16; gemm_pmc = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) per kernel, "
                "in situ: separate rocprofv3 --pmc passes around the C host running THIS workload (profiles/r03_pmc_encoder.json, "
                "scripts/gpu_r03_pmc_encoder.sh); all_encoder_side_kernels = the same ratio over log-mel, norms, attention and GEMMs together",
    }

    prof = model.profile_decode(enc, L, stride=stride)
    sampled_t = [t for t in range(L) if t % stride == stride // 2]
    rows = B * cfg.n_channels
    n_sampled = len(sampled_t)
    span = prof["unsampled_span"]
    kern = {k: v for k, v in prof.items() if k != "unsampled_span" and v["launches"] > 0}
    # An event pair costs stream time of its own, so per-launch brackets over-read.  Calibrate against the true step
    # time (one bracket around the stride-1 un-bracketed steps after every sampled step): the per-bracket overhead
    # is (sum of all brackets of a sampled step - true step time) / launches per step.
    step_true_ms = span["ms_total"] / max(1, span["launches"] * (stride - 1))
    launches_per_step = sum(v["launches"] for v in kern.values()) / n_sampled
    pair_ms = max(0.0, (sum(v["ms_total"] for v in kern.values()) / n_sampled - step_true_ms) / launches_per_step)
    us = {k: 1e3 * max(0.0, v["ms_total"] / v["launches"] - pair_ms) for k, v in kern.items()}       # per launch
    per_step = {k: v["launches"] / n_sampled for k, v in kern.items()}                                 # launches per step
    step_us = sum(us[k] * per_step[k] for k in kern)

    sa_bytes = sum(self_attn_bytes(cfg, rows, t) for t in sampled_t) / n_sampled
    ca_bytes = cross_attn_bytes(cfg, B)
    wb = decoder_weight_bytes(cfg)
    # a layer's self- and cross-attention are one launch at this shape (dec_attn_pair_kernel): its bytes are both K/V streams
    paired = "attn_pair" in kern
    stepk = "step_layers" in kern          # YMT3_STEP_KERNEL=1: a step's six layers are one launch (all K/V streams + every weight but layer 0's QKV)
    attn_name = "step_layers" if stepk else ("attn_pair" if paired else "self_attn")
    attn_bytes = sa_bytes + ca_bytes if paired else sa_bytes
    if stepk:
        attn_bytes = cfg.n_dec_layers * (sa_bytes + ca_bytes) + sum(v for k, v in wb.items() if k != "lm_head_gemm") * cfg.n_dec_layers + wb["lm_head_gemm"] - wb["qkv_cache_gemm"]
    assert kern[attn_name]["launches"] == n_sampled * (1 if stepk else cfg.n_dec_layers), (kern[attn_name], n_sampled)

    def hbm(bytes_per_launch, name):
        gbs = bytes_per_launch / (us[name] * 1e-6) / 1e9
        return {"achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "avg_launch_us": us[name], "algorithmic_bytes_per_launch": bytes_per_launch,
                "share_of_step": us[name] * per_step[name] / step_us}

    sa = hbm(attn_bytes, attn_name)
    shares = {k: us[k] * per_step[k] / step_us for k in kern}
    top = max(shares, key=shares.get)
    kernel_label = {"self_attn": "dec_attn_kernel<true> (decoder self-attention over the KV cache)",
                    "attn_pair": "dec_attn_pair_kernel (a decoder layer's self-attention over the KV cache + cross-attention over the encoder K/V, one launch)",
                    "step_layers": "dec_step_kernel (all six decoder layers of a step as one launch: YMT3_STEP_KERNEL=1)"}
    out["roofline"] = {
        "kernel": kernel_label.get(top, top),
        "bound": "hbm", "achieved": sa["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sa["frac"],
        "traffic": None if stepk else pmc_traffic(cfg, B, L, paired),
        "avg_launch_us": sa["avg_launch_us"], "launches_timed": kern[attn_name]["launches"], "event_pair_overhead_us": 1e3 * pair_ms,
        "eager_step_us": 1e3 * step_true_ms, "share_of_step": sa["share_of_step"],
        "algorithmic_bytes_per_launch": attn_bytes,
        "note": "top kernel by aggregate share of the decode step; bytes = rows*heads*(t+1)*64*2B*2 (self K and V) averaged over sampled positions "
                "t = stride/2, 3*stride/2, ... "
                + ("+ segments*heads*frames*64*2B*2 (cross K and V); " if paired else "; ") +
                "duration = HIP events around each sampled launch on the launch stream minus the per-bracket "
                "overhead calibrated against un-bracketed steps; traffic = FETCH_SIZE*2 + WRITE_SIZE per launch from "
                "the committed rocprofv3 --pmc passes under profiles/ (separate passes per counter)",
    }
    total_bytes = decode_bytes_per_batch(cfg, B, L)
    path_gbs = total_bytes / sec_per_batch / 1e9
    out["roofline_path"] = {
        "bound": "hbm", "achieved": path_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": path_gbs / HBM_PEAK_GBS,
        "algorithmic_bytes_per_batch": total_bytes,
        "note": "whole hot path: (decoder weights + cross K/V per step + self K/V cache at each position, summed over the decode) / "
                "measured seconds per batch (graph replay, the timed region above); front-end and encoder (MFMA-bound, < 1 % of a batch) add no bytes here",
    }
    gemm_names = [k for k in kern if k.endswith("_gemm") or k == "gemm_chain"]
    gemm_us = sum(us[k] * per_step[k] for k in gemm_names)
    gemm_bytes = sum(wb.get(k, 0) * per_step[k] for k in gemm_names)
    if "gemm_chain" in kern:       # one launch = cross O + FFN-in + FFN-out + the next layer's QKV projection (lm_head after the last layer)
        n_chain = per_step["gemm_chain"]
        gemm_bytes += n_chain * (wb["cross_o_gemm"] + wb["ffn_wi_gemm"] + wb["ffn_wo_gemm"]) + (n_chain - 1) * wb["qkv_cache_gemm"] + wb["lm_head_gemm"]
    out["roofline_kernels"] = {
        # (None when the cross-attention runs inside the attention-pair launch: the roofline entry above covers both streams)
        "cross_attn": hbm(ca_bytes + (wb["cross_q"] if "cross_q_gemm" not in kern else 0), "cross_attn") if "cross_attn" in kern else None,
        "dec_gemm_family": {"share_of_step": gemm_us / step_us, "launches_per_step": sum(per_step[k] for k in gemm_names),
                            "weight_bytes_per_step": gemm_bytes, "achieved": gemm_bytes / (gemm_us * 1e-6) / 1e9,
                            "frac": gemm_bytes / (gemm_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "unit": "GB/s",
                            "note": "latency-bound skinny GEMMs (64 rows; a gemm_chain launch holds four of them): weight bytes / time, against the HBM peak"},
        "launches_per_step": launches_per_step,
    }
    out["decode_step_breakdown_us"] = {k: round(us[k] * per_step[k], 2) for k in kern}
    return out


def main(argv=None):
    args = parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher.  The parent never touches the GPU (no HIP call before or
        # after this point); every rank is a fresh child process with RANK / LOCAL_RANK / WORLD_SIZE set.
        from yourmt3_amd.dist import launch_local_ranks
        cmd = [sys.executable, os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
        sys.exit(launch_local_ranks(args.gpus, cmd))
    if env_world is not None and int(env_world) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: launch one rank per GPU "
                 f"(python bench.py --gpus N starts them itself)")
    run(args)


if __name__ == "__main__":
    main()
