"""Build the HIP library in-tree: hipcc --offload-arch=gfx950, one object per kernel file, then one
shared object yourmt3_amd/libymt3_hip.so (git-ignored; it travels to the GPU box with the snapshot).
Cross-compiles without a GPU.  `python -m yourmt3_amd.build [--force]`.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "obj")
LIB = os.path.join(HERE, "libymt3_hip.so")
SOURCES = ["runtime.hip", "frontend.hip", "gemm.hip", "norm.hip", "enc_attn.hip", "decode.hip", "dec_chain.hip", "dec_step.hip", "moe.hip", "moe_chain.hip", "mc_cross_attn.hip", "ingest.hip"]
HEADERS = ["common.h", "kernels.h", "dec_chain_body.h", os.path.join("..", "..", "include", "ymt3.h")]
# -amdgpu-kernarg-preload-count: leading scalar kernel arguments arrive in SGPRs with the dispatch (gfx950) instead of through a
# scalar load at the head of the kernel; the decode-step kernels put their operand pointers there (decode.hip)
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-fno-gpu-rdc",
         "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, flags=None, lib: str = LIB, obj_dir: str = OBJ) -> str:
    """`flags` / `lib` / `obj_dir`: A/B builds of the same sources under other compiler flags (scripts/), selected at run
    time with YMT3_LIB; the product build uses the defaults."""
    flags = FLAGS if flags is None else flags
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + flags + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
            if r.stderr.strip() and verbose:
                print(r.stderr)
        return o

    with ThreadPoolExecutor(max_workers=min(6, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return lib


def build_tools(force: bool = False) -> str:
    """tools/ymt3_run: the torch-free C++ host over the C ABI (used by scripts/gpu_pmc.sh).  Rebuilt whenever the
    header or the library changes -- its `ymt3_config` must match include/ymt3.h."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tools", "ymt3_run.cpp")
    out = os.path.join(root, "tools", "ymt3_run")
    if not os.path.exists(src):
        return ""
    deps = [src, os.path.join(root, "include", "ymt3.h"), LIB]
    if force or _stale(out, deps):
        cmd = [_hipcc(), "-O2", src, "-I" + os.path.join(root, "include"), "-L" + HERE, "-lymt3_hip",
               "-Wl,-rpath,$ORIGIN/../yourmt3_amd", "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"tools/ymt3_run failed to build:\n{r.stdout}\n{r.stderr}")
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_tools(force="--force" in sys.argv))
