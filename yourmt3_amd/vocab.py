"""MIDI-event token vocabulary (host side, integer work).

Nothing in /root/reference defines it (README + LICENSE only); BASELINE.json `north_star` asks to keep
"YourMT3's MIDI-event token vocabulary", which SURVEY section 9 recalls (UNVERIFIED) as MT3-style: specials
PAD=0 / EOS=1 / UNK=2, then contiguous event ranges shift, pitch, velocity, tie, program, drum.  This
module states that codec explicitly; ids above `Codec.size` are unused padding of the 1536-wide head.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

PAD, EOS, UNK = 0, 1, 2
NUM_SPECIAL = 3


@dataclass(frozen=True)
class Event:
    type: str
    value: int


@dataclass(frozen=True)
class EventRange:
    type: str
    min_value: int
    max_value: int

    @property
    def size(self) -> int:
        return self.max_value - self.min_value + 1


class Codec:
    """Bijective map  event <-> token id  over contiguous ranges, specials first."""

    def __init__(self, max_shift_steps: int = 206, steps_per_second: int = 100):
        self.max_shift_steps = max_shift_steps
        self.steps_per_second = steps_per_second
        self.ranges: List[EventRange] = [
            EventRange("shift", 1, max_shift_steps),     # advance time by n steps of 10 ms
            EventRange("pitch", 0, 127),
            EventRange("velocity", 0, 1),                # 0 = the following pitches are offsets, 1 = onsets
            EventRange("tie", 0, 0),                     # closes the tie section that opens every segment
            EventRange("program", 0, 129),               # 0-127 GM, 128 = drums, 129 = singing voice
            EventRange("drum", 0, 127),
        ]
        self._offset = {}
        off = NUM_SPECIAL
        for r in self.ranges:
            self._offset[r.type] = (off, r)
            off += r.size
        self.size = off

    def encode(self, ev: Event) -> int:
        off, r = self._offset[ev.type]
        if not r.min_value <= ev.value <= r.max_value:
            raise ValueError(f"{ev} outside [{r.min_value}, {r.max_value}]")
        return off + ev.value - r.min_value

    def decode(self, token: int) -> Event:
        if token < NUM_SPECIAL or token >= self.size:
            return Event("special", int(token))
        for t, (off, r) in self._offset.items():
            if off <= token < off + r.size:
                return Event(t, r.min_value + token - off)
        raise AssertionError

    def range_of(self, type_: str) -> Tuple[int, int]:
        off, r = self._offset[type_]
        return off, off + r.size
