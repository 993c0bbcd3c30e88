"""Data-parallel sharding of segment batches: one process per GPU, segments are
independent units, the only collective is an all-gather of int32 token ids (<= ~0.5 MB per rank, so
latency-bound: one single-shot all-gather per batch, RCCL over xGMI; `nccl` IS RCCL on ROCm).
Works on CPU with gloo for the world_size-2 tests.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def launch_local_ranks(n_ranks: int, cmd: Sequence[str], extra_env: Optional[Dict[str, str]] = None,
                       timeout: Optional[float] = None) -> int:
    """Start `n_ranks` fresh child processes of `cmd` on this node, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT set (what torch.distributed.run would set), wait for all of them and return the
    first non-zero exit code (0 if every rank succeeded).  The caller must not have initialised HIP: children are new
    processes (never an exec of a process that touched the GPU), and they inherit stdout, so rank 0's one JSON line is
    the launcher's output.  When a rank fails, the others are terminated by PID (they would otherwise wait in a
    collective for ever)."""
    if n_ranks < 1:
        raise ValueError("n_ranks must be >= 1")
    port = _free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_ranks), "LOCAL_WORLD_SIZE": str(n_ranks),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen(list(cmd), env=env))
    deadline = None if timeout is None else time.monotonic() + timeout
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
        if rc != 0 or (deadline is not None and time.monotonic() > deadline):
            if rc == 0:
                rc = 124
                print(f"launch_local_ranks: timeout after {timeout} s", file=sys.stderr)
            for p in live:
                p.terminate()
            for p in live:
                try:
                    p.wait(10)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.05)
    return rc


def init_distributed(n_gpus: int = 1) -> Tuple[int, int, int]:
    """Read RANK / WORLD_SIZE / LOCAL_RANK (set by torch.distributed.run or launch_local_ranks) and join the process
    group.  `n_gpus` is what the caller asked for: a mismatch with WORLD_SIZE is an error, never a silent 1-rank run."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if n_gpus != world:
        raise RuntimeError(f"asked for {n_gpus} ranks but WORLD_SIZE={world}: start one process per GPU "
                           "(yourmt3_amd.dist.launch_local_ranks or torch.distributed.run)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # YMT3_DIST_BACKEND=gloo lets several ranks rehearse on ONE GPU (RCCL refuses duplicate devices)
        backend = os.environ.get("YMT3_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            n_dev = max(1, torch.cuda.device_count())
            # (Several ranks rehearsing on ONE GPU can starve each other's merged decode kernels, which need every CU for their own workgroups
            # while they run: a stage then gives up after 1 s, the call is re-run through the separate launches -- same ids -- and the
            # handle stays on them: include/ymt3.h, ymt3_set_abort_recovery.  Nothing to arrange here.)
            local_rank %= n_dev
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def shard_range(n_segments: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block split: rank r owns [r*ceil(n/W), min((r+1)*ceil(n/W), n))."""
    per = -(-n_segments // world)
    lo = min(rank * per, n_segments)
    return lo, min(lo + per, n_segments)


def shard_sizes(n_segments: int, world: int) -> List[int]:
    return [shard_range(n_segments, r, world)[1] - shard_range(n_segments, r, world)[0] for r in range(world)]


def all_gather_tokens(tokens: torch.Tensor, world: int, n_segments: int | None = None, always_collective: bool = False) -> torch.Tensor:
    """(b_local, K, L) int32 per rank -> (n_segments, K, L) on every rank, in segment order.

    Equal shards use one all_gather_into_tensor.  Ragged shards (the tail rank has fewer segments)
    are padded to the largest shard, gathered, and the padding rows dropped.  A single rank returns its tokens as they are
    unless `always_collective` (a test's way to run the collective itself -- RCCL on one GPU -- through this code path).
    """
    if world == 1 and not always_collective:
        return tokens
    b, K, L = tokens.shape
    if n_segments is None:
        n_segments = b * world
    sizes = shard_sizes(n_segments, world)
    bmax = max(sizes)
    if b < bmax:
        pad = torch.zeros(bmax - b, K, L, dtype=tokens.dtype, device=tokens.device)
        tokens = torch.cat([tokens, pad], 0)
    out = torch.empty(world * bmax, K, L, dtype=tokens.dtype, device=tokens.device)
    if dist.get_backend() == "gloo" and tokens.is_cuda:      # rehearsal on one GPU: gloo gathers through the host
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, tokens.contiguous().cpu())
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, tokens.contiguous())
    if all(s == bmax for s in sizes):
        return out
    return torch.cat([out[r * bmax:r * bmax + sizes[r]] for r in range(world)], 0)


def gather_floats(values: Sequence[float], world: int, device: Optional[torch.device] = None) -> List[List[float]]:
    """Every rank contributes the same number of floats (timings); every rank gets them back as [rank][i].  Used by bench.py so that
    rank 0's one JSON line carries every rank's own times (a slow rank or a slow collective is then visible in the line itself)."""
    mine = torch.tensor(list(values), dtype=torch.float64)
    if world == 1:
        return [mine.tolist()]
    on_gpu = dist.get_backend() == "nccl"
    if on_gpu:
        mine = mine.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    out = torch.empty(world * mine.numel(), dtype=torch.float64, device=mine.device)
    dist.all_gather_into_tensor(out, mine)
    return out.view(world, -1).cpu().tolist()
