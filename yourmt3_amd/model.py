"""Host-side mirror of the model object on the reference's inference path.

SURVEY.md section 9 (unverified recollection; nothing in /root/reference to cite) has upstream
`model/ymt3.py` exposing `inference(x, task_tokens, max_token_length)` and
`inference_file(bsz, audio_segments)` returning a list of (B, K, L) int arrays.  `YourMT3` keeps
those names and shapes; every tensor operation is a call through the C ABI (include/ymt3.h) into
the gfx950 kernels.  torch is used for device buffers and the current HIP stream only.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from .config import YMT3Config, to_c
from .tables import derived_tables
from .weights import make_weights, pack_blob


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class YourMT3:
    def __init__(self, cfg: YMT3Config, weights: Optional[Dict[str, torch.Tensor]] = None, *, seed: int = 1234,
                 device: int = 0, max_batch: int = 64):
        if not torch.cuda.is_available():
            raise _lib.YMT3Error("no GPU visible: the MI355X HIP path cannot run (there is no CPU fallback)")
        self.cfg = cfg
        self.device = torch.device("cuda", device)
        self.max_batch = int(max_batch)
        self._lib = _lib.load()
        self.weights = weights if weights is not None else make_weights(cfg, seed)
        blob = pack_blob({**self.weights, **derived_tables(self.weights, cfg)})
        self._handle = ctypes.c_void_p()
        ccfg = to_c(cfg, self.max_batch)
        # `blob` is immutable bytes: c_char_p points at its buffer (no second ~91 MB host copy); ymt3_create only reads it
        _lib.check(self._lib.ymt3_create(ctypes.byref(ccfg), ctypes.c_char_p(blob), len(blob), device, ctypes.byref(self._handle)))

    def close(self):
        if getattr(self, "_handle", None) and self._handle.value:
            self._lib.ymt3_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    @property
    def device_bytes(self) -> int:
        return int(self._lib.ymt3_device_bytes(self._handle))

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _audio2d(self, audio: torch.Tensor) -> torch.Tensor:
        if audio.dim() == 3:                       # (B, 1, S) as upstream slices it
            audio = audio[:, 0, :]
        if audio.shape[-1] != self.cfg.segment_samples:
            raise ValueError(f"segments must have {self.cfg.segment_samples} samples, got {audio.shape[-1]}")
        if audio.shape[0] > self.max_batch:
            raise ValueError(f"batch {audio.shape[0]} exceeds max_batch {self.max_batch}")
        return audio.to(self.device, torch.float32).contiguous()

    @property
    def last_decode_steps(self) -> int:
        """Decoder steps the last decode / inference call launched (fewer than asked for after an early stop)."""
        return int(self._lib.ymt3_last_decode_steps(self._handle))

    @property
    def merged_fallbacks(self) -> int:
        """How often this handle left the merged decode kernels for the separate launches after one gave up waiting (0 or 1)."""
        return int(self._lib.ymt3_merged_fallbacks(self._handle))

    @property
    def last_decode_chains(self) -> int:
        """Concurrent row ranges the last decode call cut its batch into (include/ymt3.h: 2 for 168-256 rows of one channel, else 1)."""
        return int(self._lib.ymt3_last_decode_chains(self._handle))

    def set_abort_recovery(self, mode: int) -> None:
        """1 (default): decode calls verify at their end that no merged kernel gave up and re-run through the separate launches if
        one did; 0: fully asynchronous calls, an aborted call's ids are INT32_MIN and the next call switches over (include/ymt3.h)."""
        _lib.check(self._lib.ymt3_set_abort_recovery(self._handle, int(mode)))

    def set_early_stop(self, interval: int) -> None:
        """Check every `interval` steps whether all rows have emitted EOS and stop decoding once they have (0 = off)."""
        _lib.check(self._lib.ymt3_set_early_stop(self._handle, int(interval)))

    # ------------------------------------------------------------------ stages (C ABI, 1:1)
    def ingest(self, pcm: torch.Tensor, sample_rate: int) -> torch.Tensor:
        """(n_frames, n_channels) or (n_frames,) int16 / float32 PCM at `sample_rate` -> (n_seg, 1, S) float32 mono
        segments at cfg.sample_rate on the device (mix, resample, slice and zero-pad in one kernel)."""
        if pcm.dim() == 1:
            pcm = pcm[:, None]
        if pcm.dim() != 2:
            raise ValueError("pcm must be (n_frames, n_channels)")
        if pcm.dtype == torch.int16:
            fmt = 0
        elif pcm.dtype == torch.float32:
            fmt = 1
        else:
            raise ValueError("pcm must be int16 or float32")
        pcm = pcm.to(self.device).contiguous()
        n_frames, n_ch = int(pcm.shape[0]), int(pcm.shape[1])
        n_out, n_seg = ctypes.c_int64(0), ctypes.c_int(0)
        _lib.check(self._lib.ymt3_ingest_plan(self._handle, n_frames, int(sample_rate), ctypes.byref(n_out), ctypes.byref(n_seg)))
        segs = torch.empty(n_seg.value, 1, self.cfg.segment_samples, device=self.device, dtype=torch.float32)
        _lib.check(self._lib.ymt3_ingest(self._handle, _ptr(pcm) if n_frames else None, fmt, n_frames, n_ch, int(sample_rate),
                                         _ptr(segs), n_seg.value, self._stream()))
        self.last_ingest_samples = int(n_out.value)
        return segs

    def logmel(self, audio: torch.Tensor) -> torch.Tensor:
        a = self._audio2d(audio)
        B = a.shape[0]
        mel = torch.empty(B, self.cfg.n_frames, self.cfg.n_mels, device=self.device, dtype=torch.float32)
        _lib.check(self._lib.ymt3_logmel(self._handle, _ptr(a), B, _ptr(mel), self._stream()))
        return mel

    def encode(self, mel: torch.Tensor) -> torch.Tensor:
        mel = mel.to(self.device, torch.float32).contiguous()
        B = mel.shape[0]
        enc = torch.empty(B, self.cfg.n_frames, self.cfg.d_model, device=self.device, dtype=torch.bfloat16)
        _lib.check(self._lib.ymt3_encode(self._handle, _ptr(mel), B, _ptr(enc), self._stream()))
        return enc

    def decode(self, enc: torch.Tensor, n_steps: Optional[int] = None, forced: Optional[torch.Tensor] = None,
               return_logits: bool = False):
        cfg = self.cfg
        n_steps = int(n_steps or cfg.max_decode_len)
        enc = enc.to(self.device, torch.bfloat16).contiguous()
        B = enc.shape[0]
        tokens = torch.empty(B, cfg.n_channels, n_steps, device=self.device, dtype=torch.int32)
        f = forced.to(self.device, torch.int32).contiguous() if forced is not None else None
        if f is not None and tuple(f.shape) != (B, cfg.n_channels, n_steps):
            raise ValueError("forced must be (B, n_channels, n_steps)")
        lg = torch.empty(B, cfg.n_channels, n_steps, cfg.vocab, device=self.device, dtype=torch.float32) if return_logits else None
        _lib.check(self._lib.ymt3_decode_greedy(self._handle, _ptr(enc), B, n_steps, _ptr(tokens), _ptr(f), _ptr(lg), self._stream()))
        return (tokens, lg) if return_logits else tokens

    # ------------------------------------------------------------------ reference-shaped API
    def inference(self, audio: torch.Tensor, task_tokens=None, max_token_length: Optional[int] = None) -> torch.Tensor:
        """(B, 1, S) or (B, S) audio -> (B, K, L) int32 token ids: the whole hot path, one C call."""
        a = self._audio2d(audio)
        B = a.shape[0]
        L = int(max_token_length or self.cfg.max_decode_len)
        tokens = torch.empty(B, self.cfg.n_channels, L, device=self.device, dtype=torch.int32)
        _lib.check(self._lib.ymt3_transcribe_segments(self._handle, _ptr(a), B, L, _ptr(tokens), self._stream()))
        return tokens

    def inference_stream(self, audio_segments: torch.Tensor, max_token_length: Optional[int] = None, slots: int = 0,
                         interval: int = 8) -> torch.Tensor:
        """(N, 1, S) or (N, S) audio, any N -> (N, K, L) int32 ids with continuous batching: `slots` decoder slots are
        refilled from the queue as segments emit EOS (needs eos_id >= 0 to gain anything).  Ids equal inference()'s."""
        a = audio_segments[:, 0, :] if audio_segments.dim() == 3 else audio_segments
        if a.shape[-1] != self.cfg.segment_samples:
            raise ValueError(f"segments must have {self.cfg.segment_samples} samples, got {a.shape[-1]}")
        a = a.to(self.device, torch.float32).contiguous()
        N = a.shape[0]
        L = int(max_token_length or self.cfg.max_decode_len)
        tokens = torch.empty(N, self.cfg.n_channels, L, device=self.device, dtype=torch.int32)
        _lib.check(self._lib.ymt3_transcribe_stream(self._handle, _ptr(a) if N else None, N, L, _ptr(tokens) if N else None,
                                                    int(slots), int(interval), self._stream()))
        return tokens

    def inference_file(self, bsz: int, audio_segments: torch.Tensor, max_token_length: Optional[int] = None) -> List[np.ndarray]:
        """Split (N, 1, S) segments into batches of `bsz`; one (b, K, L) int array per batch."""
        bsz = min(int(bsz), self.max_batch)
        out = []
        for i in range(0, audio_segments.shape[0], bsz):
            out.append(self.inference(audio_segments[i:i + bsz], max_token_length=max_token_length).cpu().numpy())
        return out

    PROFILE_CLASSES = ["qkv_cache_gemm", "self_attn", "self_o_gemm", "cross_q_gemm", "cross_attn", "cross_o_gemm",
                       "ffn_wi_gemm", "ffn_wo_gemm", "lm_head_gemm", "argmax_embed", "unsampled_span", "gemm_chain", "attn_pair", "step_layers"]

    def profile_decode(self, enc: torch.Tensor, n_steps: int, stride: int = 16) -> Dict[str, dict]:
        """Eager decode with HIP events around each kernel of every `stride`-th step (include/ymt3.h)."""
        enc = enc.to(self.device, torch.bfloat16).contiguous()
        B = enc.shape[0]
        tokens = torch.empty(B, self.cfg.n_channels, n_steps, device=self.device, dtype=torch.int32)
        ms = (ctypes.c_float * 16)()
        cnt = (ctypes.c_int32 * 16)()
        _lib.check(self._lib.ymt3_profile_decode(self._handle, _ptr(enc), B, n_steps, stride, _ptr(tokens), ms, cnt, self._stream()))
        return {n: {"ms_total": float(ms[i]), "launches": int(cnt[i])} for i, n in enumerate(self.PROFILE_CLASSES)}

    def moe_trace(self, n_steps: int) -> torch.Tensor:
        """Debug hook (needs YMT3_DEBUG_HOOKS=1 at construction, MoE decoder): from now on lock-step decode calls record the router's
        choices in the returned (n_steps, n_dec_layers, max_batch * n_channels, 2) int32 device tensor (-1 where nothing was recorded)."""
        rows = self.max_batch * self.cfg.n_channels
        self._moe_trace = torch.full((n_steps, self.cfg.n_dec_layers, rows, 2), -1, device=self.device, dtype=torch.int32)
        _lib.check(self._lib.ymt3_debug_moe_trace(self._handle, _ptr(self._moe_trace), n_steps, rows))
        return self._moe_trace

    def step_stamps(self):
        """Measurement (needs YMT3_STAMP=1 at construction): per kernel of the last decode step, in launch order:
        (class name, workgroups, first entry, last entry, first exit, last exit) in microseconds from the step's start."""
        cls = np.zeros(64, np.int32); grid = np.zeros(64, np.int32); st = np.zeros((64, 4), np.uint64)
        n = ctypes.c_int(0)
        _lib.check(self._lib.ymt3_debug_step_stamps(self._handle, cls.ctypes.data, grid.ctypes.data, st.ctypes.data, ctypes.byref(n)))
        t0 = int(st[0, 0])
        return [(self.PROFILE_CLASSES[int(cls[i])], int(grid[i])) + tuple((int(v) - t0) / 100.0 for v in st[i]) for i in range(n.value)]

    def kernel_stamps(self, kernel: int, grid: int) -> np.ndarray:
        """Measurement: (grid, 2) raw 100 MHz (entry, exit) stamps of kernel `kernel` of the last decode step."""
        out = np.zeros((grid, 2), np.uint64)
        _lib.check(self._lib.ymt3_debug_kernel_stamps(self._handle, int(kernel), out.ctypes.data, int(grid)))
        return out

    def test_gemm(self, a_bf16: torch.Tensor, w_bf16: torch.Tensor) -> torch.Tensor:
        M, K = a_bf16.shape
        N = w_bf16.shape[0]
        c = torch.empty(M, N, device=self.device, dtype=torch.float32)
        _lib.check(self._lib.ymt3_test_gemm(self._handle, _ptr(a_bf16.contiguous()), _ptr(w_bf16.contiguous()), _ptr(c), M, N, K, self._stream()))
        return c
