"""ctypes binding of the C ABI in include/ymt3.h.  No fallback: if the HIP library is missing
or fails to load, importing a model raises -- the product path never routes through a CPU restatement.
"""
from __future__ import annotations

import ctypes
import os

from .config import CConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("YMT3_LIB", os.path.join(_HERE, "libymt3_hip.so"))

# every symbol include/ymt3.h declares (tests/test_cpu_host.py::test_library_builds_loads_and_exports_every_header_symbol checks the header against this list)
SYMBOLS = [
    "ymt3_abi_version", "ymt3_last_error", "ymt3_create", "ymt3_destroy", "ymt3_device_bytes",
    "ymt3_logmel", "ymt3_encode", "ymt3_decode_greedy", "ymt3_transcribe_segments", "ymt3_test_gemm", "ymt3_profile_decode", "ymt3_debug_decode_start", "ymt3_debug_force_stage_abort", "ymt3_set_early_stop", "ymt3_last_decode_steps",
    "ymt3_ingest_plan", "ymt3_ingest", "ymt3_transcribe_stream", "ymt3_debug_step_stamps", "ymt3_debug_kernel_stamps",
    "ymt3_set_abort_recovery", "ymt3_merged_fallbacks", "ymt3_debug_moe_trace", "ymt3_last_decode_chains",
]

_lib = None


class YMT3Error(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise YMT3Error(
            f"{LIB_PATH} not found: build it with `python -m yourmt3_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the hot path.")
    # The process must hold ONE HIP runtime.  torch ships its own libamdhip64.so (SONAME libamdhip64.so.7)
    # and device pointers / streams are shared with it, so torch has to be loaded first: our NEEDED
    # libamdhip64.so.7 then binds to torch's copy.  Loaded the other way round the process ends up with
    # two runtimes and the second one sees no device.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    lib.ymt3_abi_version.restype = i32
    lib.ymt3_last_error.restype = ctypes.c_char_p
    lib.ymt3_create.argtypes = [ctypes.POINTER(CConfig), vp, sz, i32, ctypes.POINTER(vp)]
    lib.ymt3_create.restype = i32
    lib.ymt3_destroy.argtypes = [vp]
    lib.ymt3_destroy.restype = None
    lib.ymt3_device_bytes.argtypes = [vp]
    lib.ymt3_device_bytes.restype = sz
    lib.ymt3_logmel.argtypes = [vp, vp, i32, vp, vp]
    lib.ymt3_encode.argtypes = [vp, vp, i32, vp, vp]
    lib.ymt3_decode_greedy.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp]
    lib.ymt3_transcribe_segments.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.ymt3_test_gemm.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp]
    lib.ymt3_profile_decode.argtypes = [vp, vp, i32, i32, i32, vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32), vp]
    lib.ymt3_profile_decode.restype = i32
    lib.ymt3_debug_decode_start.argtypes = [vp, i32]
    lib.ymt3_debug_decode_start.restype = i32
    lib.ymt3_debug_force_stage_abort.argtypes = [vp]
    lib.ymt3_debug_force_stage_abort.restype = i32
    lib.ymt3_last_decode_steps.argtypes = [vp]
    lib.ymt3_last_decode_steps.restype = i32
    lib.ymt3_set_early_stop.argtypes = [vp, i32]
    lib.ymt3_set_early_stop.restype = i32
    lib.ymt3_ingest_plan.argtypes = [vp, ctypes.c_int64, i32, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int)]
    lib.ymt3_ingest_plan.restype = i32
    lib.ymt3_ingest.argtypes = [vp, vp, i32, ctypes.c_int64, i32, i32, vp, i32, vp]
    lib.ymt3_ingest.restype = i32
    lib.ymt3_transcribe_stream.argtypes = [vp, vp, i32, i32, vp, i32, i32, vp]
    lib.ymt3_transcribe_stream.restype = i32
    lib.ymt3_debug_step_stamps.argtypes = [vp, vp, vp, vp, ctypes.POINTER(ctypes.c_int)]
    lib.ymt3_debug_step_stamps.restype = i32
    lib.ymt3_debug_kernel_stamps.argtypes = [vp, i32, vp, i32]
    lib.ymt3_debug_kernel_stamps.restype = i32
    lib.ymt3_set_abort_recovery.argtypes = [vp, i32]
    lib.ymt3_set_abort_recovery.restype = i32
    lib.ymt3_merged_fallbacks.argtypes = [vp]
    lib.ymt3_merged_fallbacks.restype = i32
    lib.ymt3_last_decode_chains.argtypes = [vp]
    lib.ymt3_last_decode_chains.restype = i32
    lib.ymt3_debug_moe_trace.argtypes = [vp, vp, i32, i32]
    lib.ymt3_debug_moe_trace.restype = i32
    for n in ("ymt3_logmel", "ymt3_encode", "ymt3_decode_greedy", "ymt3_transcribe_segments", "ymt3_test_gemm"):
        getattr(lib, n).restype = i32
    if lib.ymt3_abi_version() != 3:
        raise YMT3Error("libymt3_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise YMT3Error(f"ymt3 error {rc}: {load().ymt3_last_error().decode(errors='replace')}")
