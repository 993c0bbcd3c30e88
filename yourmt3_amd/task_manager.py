"""TaskManager: token ids <-> note events <-> notes (host, integer state machines).

Kept API names (BASELINE.json north_star; signatures per SURVEY section 9, UNVERIFIED -- the reference tree has
no code): `TaskManager(task_name, max_shift_steps)`, `.tokenizer`, `.num_decoding_channels`,
`.max_note_token_length`, `.detokenize_list_batches(list_batch_token_arrays, list_start_sec, return_events)`.

Token grammar of one segment (MT3 style):
    [program p, pitch k]*  TIE     notes still sounding from the previous segment (tie section)
    then events in time order:  SHIFT n (time += n*10 ms) | VELOCITY v | PROGRAM p | PITCH k | DRUM k
    EOS, then PAD.
VELOCITY 1 makes the following pitches onsets, VELOCITY 0 offsets; drums have onsets only.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .vocab import Codec, Event, EOS, PAD, UNK

DRUM_PROGRAM = 128
DRUM_NOTE_SEC = 0.01          # drums carry no offset: fixed nominal duration


@dataclass(frozen=True, order=True)
class NoteEvent:
    time: float
    is_drum: bool
    program: int
    velocity: int             # 1 onset, 0 offset
    pitch: int


@dataclass(frozen=True, order=True)
class Note:
    onset: float
    offset: float
    is_drum: bool
    program: int
    pitch: int
    velocity: int = 100


# instrument classes of the 13-channel decoder (MT3 "FULL_PLUS" grouping + singing + drums)
MC13_GROUPS: List[Tuple[str, Sequence[int]]] = [
    ("piano", range(0, 8)), ("chromatic_percussion", range(8, 16)), ("organ", range(16, 24)),
    ("guitar", range(24, 32)), ("bass", range(32, 40)), ("strings", range(40, 56)), ("brass", range(56, 64)),
    ("reed", range(64, 72)), ("pipe", range(72, 80)), ("synth_lead", range(80, 88)), ("synth_pad", range(88, 96)),
    ("singing", (129,)), ("drums", (128,)),
]

TASKS: Dict[str, dict] = {
    "mt3_full_plus": {"channels": 1, "max_tokens": 1024},
    "mc13_full_plus_256": {"channels": 13, "max_tokens": 256},
}


class NoteEventTokenizer:
    def __init__(self, codec: Codec):
        self.codec = codec

    # ---------------------------------------------------------------- notes -> tokens (for tests / data)
    def encode_segment(self, events: Sequence[NoteEvent], tie_notes: Sequence[Tuple[int, int]], start_sec: float,
                       max_len: Optional[int] = None) -> List[int]:
        c = self.codec
        toks: List[int] = []
        for prog, pitch in sorted(tie_notes):
            toks += [c.encode(Event("program", prog)), c.encode(Event("pitch", pitch))]
        toks.append(c.encode(Event("tie", 0)))
        cur_step, cur_vel, cur_prog = 0, None, None
        for ev in sorted(events):
            step = int(round((ev.time - start_sec) * c.steps_per_second))
            d = step - cur_step
            while d > 0:
                n = min(d, c.max_shift_steps)
                toks.append(c.encode(Event("shift", n)))
                d -= n
            cur_step = max(cur_step, step)
            if ev.is_drum:
                if cur_vel != 1:
                    toks.append(c.encode(Event("velocity", 1)))
                    cur_vel = 1
                toks.append(c.encode(Event("drum", ev.pitch)))
                continue
            if ev.velocity != cur_vel:
                toks.append(c.encode(Event("velocity", ev.velocity)))
                cur_vel = ev.velocity
            if ev.program != cur_prog:
                toks.append(c.encode(Event("program", ev.program)))
                cur_prog = ev.program
            toks.append(c.encode(Event("pitch", ev.pitch)))
        toks.append(EOS)
        if max_len is not None:
            if len(toks) > max_len:
                raise ValueError(f"segment needs {len(toks)} tokens > {max_len}")
            toks += [PAD] * (max_len - len(toks))
        return toks

    # ---------------------------------------------------------------- tokens -> note events
    def decode_segment(self, tokens: Iterable[int], start_sec: float):
        """-> (events, tie_notes, n_invalid).  Stops at EOS/PAD; malformed tokens are counted, not fatal."""
        c = self.codec
        events: List[NoteEvent] = []
        ties: List[Tuple[int, int]] = []
        in_tie, step, vel, prog, bad = True, 0, 1, 0, 0
        for tk in tokens:
            tk = int(tk)
            if tk in (EOS, PAD):
                break
            ev = c.decode(tk)
            if ev.type == "special":                       # UNK or an id beyond the codec
                bad += 1
            elif ev.type == "tie":
                in_tie = False
            elif ev.type == "shift":
                in_tie = False
                step += ev.value
            elif ev.type == "velocity":
                vel = ev.value
            elif ev.type == "program":
                prog = ev.value
            elif ev.type == "pitch":
                if prog == DRUM_PROGRAM:                    # a pitch under the drum program is a drum hit: no ties, no offsets
                    if in_tie:
                        bad += 1
                    elif vel:
                        events.append(NoteEvent(start_sec + step / c.steps_per_second, True, DRUM_PROGRAM, 1, ev.value))
                elif in_tie:
                    ties.append((prog, ev.value))
                else:
                    events.append(NoteEvent(start_sec + step / c.steps_per_second, False, prog, vel, ev.value))
            elif ev.type == "drum":
                if in_tie:
                    bad += 1
                else:
                    events.append(NoteEvent(start_sec + step / c.steps_per_second, True, DRUM_PROGRAM, 1, ev.value))
        return events, ties, bad


def note_events_to_notes(segments: Sequence[Tuple[float, List[NoteEvent], List[Tuple[int, int]]]], end_sec: float) -> List[Note]:
    """Merge per-segment (start_sec, events, tie_notes) into notes.

    A note sounding at a segment boundary stays open only if the next segment's tie section lists it;
    otherwise it is closed at that segment's start.  Offsets without an onset are dropped; a repeated
    onset re-triggers (closes the old note at the new onset); a drum hit repeated at the same time and
    pitch (the model emitting a token twice) is one hit.
    """
    active: Dict[Tuple[int, int], float] = {}
    notes: List[Note] = []
    drum_hits = set()
    for start, events, ties in sorted(segments, key=lambda s: s[0]):
        tie_set = set(ties)
        for key in [k for k in active if k not in tie_set]:
            on = active.pop(key)
            if start > on:
                notes.append(Note(on, start, False, key[0], key[1]))
        for ev in sorted(events):
            if ev.is_drum:
                if (ev.time, ev.pitch) not in drum_hits:
                    drum_hits.add((ev.time, ev.pitch))
                    notes.append(Note(ev.time, ev.time + DRUM_NOTE_SEC, True, DRUM_PROGRAM, ev.pitch))
                continue
            key = (ev.program, ev.pitch)
            if ev.velocity:
                if key in active and ev.time > active[key]:
                    notes.append(Note(active[key], ev.time, False, key[0], key[1]))
                active[key] = ev.time
            elif key in active:
                on = active.pop(key)
                if ev.time > on:
                    notes.append(Note(on, ev.time, False, key[0], key[1]))
    for key, on in active.items():
        if end_sec > on:
            notes.append(Note(on, end_sec, False, key[0], key[1]))
    return sorted(notes)


class TaskManager:
    def __init__(self, task_name: str = "mt3_full_plus", max_shift_steps: int = 206, debug_mode: bool = False):
        if task_name not in TASKS:
            raise ValueError(f"unknown task {task_name!r}; known: {sorted(TASKS)}")
        self.task_name = task_name
        self.task = TASKS[task_name]
        self.codec = Codec(max_shift_steps=max_shift_steps)
        self.tokenizer = NoteEventTokenizer(self.codec)
        self.num_decoding_channels = self.task["channels"]
        self.max_note_token_length = self.task["max_tokens"]
        self.debug_mode = debug_mode

    def channel_of_program(self, program: int) -> int:
        if self.num_decoding_channels == 1:
            return 0
        for ch, (_, progs) in enumerate(MC13_GROUPS):
            if program in progs:
                return ch
        return 0

    def detokenize_list_batches(self, list_batch_token_arrays: Sequence[np.ndarray], list_start_sec: Sequence[float],
                                return_events: bool = False):
        """list of (b, L) int arrays (ONE channel: pass arr[:, ch, :]) + start time of every segment ->
        per-segment (start, events, ties); with return_events also the invalid-token count."""
        flat = np.concatenate([np.asarray(a) for a in list_batch_token_arrays], 0)
        if flat.shape[0] != len(list_start_sec):
            raise ValueError(f"{flat.shape[0]} segments but {len(list_start_sec)} start times")
        segs, bad = [], 0
        for row, start in zip(flat, list_start_sec):
            ev, ties, b = self.tokenizer.decode_segment(row, float(start))
            segs.append((float(start), ev, ties))
            bad += b
        return (segs, bad) if return_events else segs

    def tokens_to_notes(self, token_batches: Sequence[np.ndarray], start_secs: Sequence[float], end_sec: float) -> List[Note]:
        """All channels: token_batches are (b, K, L); channels are decoded independently and mixed."""
        notes: List[Note] = []
        for ch in range(self.num_decoding_channels):
            segs = self.detokenize_list_batches([np.asarray(a)[:, ch, :] for a in token_batches], start_secs)
            notes += note_events_to_notes(segs, end_sec)
        return sorted(notes)
