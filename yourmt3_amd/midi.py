"""Minimal Standard MIDI File (format 1) writer/reader for Note lists (mido / pretty_midi are absent)."""
from __future__ import annotations

import struct
from typing import Dict, List, Sequence

from .task_manager import Note, DRUM_PROGRAM

TICKS_PER_BEAT = 480
TEMPO_US = 500000                      # 120 bpm -> 960 ticks per second


def _vlq(n: int) -> bytes:
    out = [n & 0x7F]
    n >>= 7
    while n:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    return bytes(reversed(out))


def _ticks(sec: float) -> int:
    return int(round(sec * TICKS_PER_BEAT * 1e6 / TEMPO_US))


SINGING_PROGRAM = 129
SINGING_GM_PROGRAM = 53                # GM "Voice Oohs": the defined stand-in for the vocabulary's singing-voice program


def gm_program(program: int) -> int:
    """Vocabulary program -> General MIDI program number of the program-change event."""
    if program == SINGING_PROGRAM:
        return SINGING_GM_PROGRAM
    if not 0 <= program <= 127:
        raise ValueError(f"program {program} has no General MIDI number")
    return program


def _fit_melodic_channels(by_prog: Dict[int, List[Note]]) -> Dict[int, List[Note]]:
    """A Standard MIDI File has 15 melodic channels (16 minus the drum channel) and ONE program per channel at a time.
    With more than 15 melodic programs, programs of one GM instrument family (program // 8) are merged onto the family's
    lowest program present, largest families first, until 15 remain -- deterministic, and never two programs on one
    channel.  Still more than 15 distinct families (only with all 16 GM families plus singing present): error."""
    melodic = sorted(p for p in by_prog if p != DRUM_PROGRAM)
    if len(melodic) <= 15:
        return by_prog
    family = lambda p: 16 if p == SINGING_PROGRAM else p // 8
    groups: Dict[int, List[int]] = {}
    for p in melodic:
        groups.setdefault(family(p), []).append(p)
    merged = {p: p for p in melodic}
    n = len(melodic)
    for fam in sorted(groups, key=lambda f: (-len(groups[f]), f)):
        if n <= 15:
            break
        members = groups[fam]
        if len(members) > 1:
            for p in members[1:]:
                merged[p] = members[0]
            n -= len(members) - 1
    if n > 15:
        raise ValueError(f"{n} instrument families do not fit the 15 melodic MIDI channels")
    out: Dict[int, List[Note]] = {}
    for p, ns in by_prog.items():
        out.setdefault(merged.get(p, p), []).extend(ns)
    return out


def notes_to_midi_bytes(notes: Sequence[Note]) -> bytes:
    by_prog: Dict[int, List[Note]] = {}
    for n in notes:
        by_prog.setdefault(DRUM_PROGRAM if n.is_drum else n.program, []).append(n)
    tracks = [b"\x00\xff\x51\x03" + TEMPO_US.to_bytes(3, "big") + b"\x00\xff\x2f\x00"]
    by_prog = _fit_melodic_channels(by_prog)
    melodic_channels = [c for c in range(16) if c != 9]
    melodic = [p for p in sorted(by_prog) if p != DRUM_PROGRAM]
    for prog, ns in sorted(by_prog.items()):
        ch = 9 if prog == DRUM_PROGRAM else melodic_channels[melodic.index(prog)]
        ev = []
        # one voice per (channel, pitch): a note still sounding when the same pitch starts again ends at that onset, and a
        # second note starting on the same tick is dropped -- otherwise the note-ons and note-offs of the file do not pair up
        spans: Dict[int, List[List[int]]] = {}
        for n in sorted(ns, key=lambda x: (x.pitch, x.onset, x.offset)):
            on, off = _ticks(n.onset), max(_ticks(n.offset), _ticks(n.onset) + 1)
            sp = spans.setdefault(n.pitch, [])
            if sp and sp[-1][0] == on:
                sp[-1][1] = max(sp[-1][1], off)
                continue
            if sp and sp[-1][1] > on:
                sp[-1][1] = on
            sp.append([on, off, n.velocity])
        for pitch, sp in spans.items():
            for on, off, vel in sp:
                ev.append((on, 1, pitch, vel))
                ev.append((off, 0, pitch, 0))
        ev.sort(key=lambda e: (e[0], e[1]))            # offsets before onsets at the same tick
        body = bytearray()
        if prog != DRUM_PROGRAM:
            body += b"\x00" + bytes([0xC0 | ch, gm_program(prog)])
        last = 0
        for tick, on, pitch, vel in ev:
            body += _vlq(tick - last) + bytes([(0x90 if on else 0x80) | ch, pitch & 0x7F, vel & 0x7F])
            last = tick
        body += b"\x00\xff\x2f\x00"
        tracks.append(bytes(body))
    out = b"MThd" + struct.pack(">IHHH", 6, 1, len(tracks), TICKS_PER_BEAT)
    for t in tracks:
        out += b"MTrk" + struct.pack(">I", len(t)) + t
    return out


def write_midi(notes: Sequence[Note], path: str) -> str:
    with open(path, "wb") as f:
        f.write(notes_to_midi_bytes(notes))
    return path


def read_midi_notes(data: bytes) -> List[Note]:
    """Parse what notes_to_midi_bytes writes (round-trip tests): note on/off + program change only."""
    assert data[:4] == b"MThd"
    _, fmt, ntrk, tpb = struct.unpack(">IHHH", data[4:14])
    pos, notes = 14, []
    sec = lambda t: t * TEMPO_US / (tpb * 1e6)
    for _ in range(ntrk):
        assert data[pos:pos + 4] == b"MTrk"
        ln = struct.unpack(">I", data[pos + 4:pos + 8])[0]
        p, end, tick, prog, open_ = pos + 8, pos + 8 + ln, 0, 0, {}
        while p < end:
            d = 0
            while True:
                b = data[p]; p += 1
                d = (d << 7) | (b & 0x7F)
                if not b & 0x80:
                    break
            tick += d
            st = data[p]; p += 1
            if st == 0xFF:
                ml = data[p + 1]; p += 2 + ml
            elif st & 0xF0 == 0xC0:
                prog = data[p]; p += 1
            elif st & 0xF0 in (0x90, 0x80):
                pitch, vel = data[p], data[p + 1]; p += 2
                drum = (st & 0x0F) == 9
                if st & 0xF0 == 0x90 and vel > 0:
                    open_[pitch] = (tick, vel)
                elif pitch in open_:
                    on, v = open_.pop(pitch)
                    notes.append(Note(sec(on), sec(tick), drum, DRUM_PROGRAM if drum else prog, pitch, v))
        pos = end
    return sorted(notes)
