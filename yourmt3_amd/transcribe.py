"""`transcribe(model, audio_info)`: audio file or array -> MIDI file, the reference-shaped entry point
(BASELINE.json north_star; flow per SURVEY.md section 9, UNVERIFIED): load -> mono -> 16 kHz -> slice into
(n_seg, 1, 32767) -> model.inference_file(bsz, segments) -> TaskManager detokenise per channel -> notes -> MIDI."""
from __future__ import annotations

import os
from typing import Optional, Union

import numpy as np
import torch

from .audio import load_wav_pcm
from .midi import write_midi
from .task_manager import TaskManager


def transcribe(model, audio_info: Union[str, dict, np.ndarray], task_manager: Optional[TaskManager] = None, bsz: int = 8,
               output_dir: str = ".", max_token_length: Optional[int] = None, return_notes: bool = False,
               continuous: bool = False):
    """`continuous=True` decodes the file's segments through `bsz` slots with continuous batching
    (YourMT3.inference_stream: segments leave at EOS and the next ones enter) instead of fixed batches; same ids."""
    cfg = model.cfg
    if task_manager is None:
        task_manager = TaskManager("mc13_full_plus_256" if cfg.n_channels == 13 else "mt3_full_plus")
    if task_manager.num_decoding_channels != cfg.n_channels:
        raise ValueError("TaskManager channel count does not match the model's decoder")
    if isinstance(audio_info, dict):
        path = audio_info["filepath"]
        name = audio_info.get("track_name") or os.path.splitext(os.path.basename(path))[0]
        x, sr = load_wav_pcm(path)
    elif isinstance(audio_info, str):
        path, name = audio_info, os.path.splitext(os.path.basename(audio_info))[0]
        x, sr = load_wav_pcm(path)
    else:
        x, sr, name = np.asarray(audio_info, dtype=np.float32), cfg.sample_rate, "audio"
    # mono mix, resample to the model rate, slice and zero-pad on the device (C ABI: ymt3_ingest)
    segments = model.ingest(torch.from_numpy(np.ascontiguousarray(x)), sr)
    n_samples = model.last_ingest_samples
    start_secs = [i * cfg.segment_samples / cfg.sample_rate for i in range(segments.shape[0])]
    L = max_token_length or task_manager.max_note_token_length
    if continuous:
        batches = [model.inference_stream(segments, max_token_length=min(L, cfg.max_decode_len), slots=bsz).cpu().numpy()]
    else:
        batches = model.inference_file(bsz, segments, max_token_length=min(L, cfg.max_decode_len))
    notes = task_manager.tokens_to_notes(batches, start_secs, end_sec=n_samples / cfg.sample_rate)
    os.makedirs(output_dir, exist_ok=True)
    midi_path = write_midi(notes, os.path.join(output_dir, name + ".mid"))
    return (midi_path, notes) if return_notes else midi_path
