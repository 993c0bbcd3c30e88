"""Host-side (CPU) tests of the token vocabulary, TaskManager state machines, MIDI writer, audio slicing and the
`transcribe()` flow.  The reference has no tests for these; these are this build's known-answer
and round-trip tests ("pure integer, exact-parity testable; needs its own KATs")."""
import os
import random
import wave

import numpy as np
import pytest

from yourmt3_amd.audio import slice_padded_array
from host_audio import load_wav, resample
from yourmt3_amd.config import YMT3Config
from yourmt3_amd.midi import notes_to_midi_bytes, read_midi_notes, write_midi
from yourmt3_amd.task_manager import (MC13_GROUPS, Note, NoteEvent, TaskManager, note_events_to_notes, DRUM_PROGRAM)
from yourmt3_amd.transcribe import transcribe
from yourmt3_amd.vocab import Codec, Event, EOS, PAD, UNK

SEG = 32767 / 16000.0


def test_codec_is_a_bijection_and_fits_the_head():
    c = Codec()
    assert c.size == 598 <= YMT3Config().vocab
    seen = set()
    for t in range(3, c.size):
        ev = c.decode(t)
        assert c.encode(ev) == t and ev not in seen
        seen.add(ev)
    assert c.decode(PAD).type == c.decode(EOS).type == c.decode(UNK).type == c.decode(c.size).type == "special"
    assert c.decode(c.encode(Event("shift", 206))) == Event("shift", 206)
    with pytest.raises(ValueError):
        c.encode(Event("shift", 207))
    with pytest.raises(ValueError):
        c.encode(Event("pitch", 128))


def _events(notes):
    ev = []
    for n in notes:
        if n.is_drum:
            ev.append(NoteEvent(n.onset, True, DRUM_PROGRAM, 1, n.pitch))
        else:
            ev += [NoteEvent(n.onset, False, n.program, 1, n.pitch), NoteEvent(n.offset, False, n.program, 0, n.pitch)]
    return sorted(ev)


def _tokenize(tm, notes, n_seg, max_len):
    ev = _events(notes)
    rows = []
    for i in range(n_seg):
        s, e = i * SEG, (i + 1) * SEG
        ties = [(n.program, n.pitch) for n in notes if not n.is_drum and n.onset < s < n.offset]
        rows.append(tm.tokenizer.encode_segment([x for x in ev if s <= x.time < e], ties, s, max_len=max_len))
    return np.array(rows, dtype=np.int32)


def _grid(t, seg_start):            # what 10 ms quantisation relative to the segment start does to a time
    return seg_start + round((t - seg_start) * 100) / 100


def test_known_answer_token_stream():
    tm = TaskManager()
    c = tm.codec
    toks = tm.tokenizer.encode_segment(_events([Note(0.10, 0.50, False, 0, 60)]), [(40, 64)], 0.0, max_len=16)
    expect = [c.encode(Event("program", 40)), c.encode(Event("pitch", 64)), c.encode(Event("tie", 0)),
              c.encode(Event("shift", 10)), c.encode(Event("velocity", 1)), c.encode(Event("program", 0)), c.encode(Event("pitch", 60)),
              c.encode(Event("shift", 40)), c.encode(Event("velocity", 0)), c.encode(Event("pitch", 60)), EOS]
    assert toks == expect + [PAD] * 5
    ev, ties, bad = tm.tokenizer.decode_segment(toks, 0.0)
    assert ties == [(40, 64)] and bad == 0
    assert ev == [NoteEvent(0.1, False, 0, 1, 60), NoteEvent(0.5, False, 0, 0, 60)]


def test_notes_roundtrip_across_segments_with_ties_and_drums():
    rng = random.Random(7)
    tm = TaskManager()
    notes = []
    for _ in range(60):
        on = round(rng.uniform(0, 7.5), 2)
        if rng.random() < 0.2:
            notes.append(Note(on, on + 0.01, True, DRUM_PROGRAM, rng.choice([36, 38, 42])))
        else:
            notes.append(Note(on, round(on + rng.uniform(0.05, 3.0), 2), False, rng.choice([0, 24, 40, 129]), rng.randrange(30, 90)))
    # no two melodic notes of the same (program, pitch) may overlap for an exact round trip
    uniq, keep = {}, []
    for n in sorted(notes):
        k = (n.program, n.pitch)
        if n.is_drum or k not in uniq or uniq[k] < n.onset - 0.02:
            keep.append(n)
            if not n.is_drum:
                uniq[k] = n.offset
    n_seg = 5
    toks = _tokenize(tm, keep, n_seg, 512)
    got = tm.tokens_to_notes([toks[:2, None, :], toks[2:, None, :]], [i * SEG for i in range(n_seg)], end_sec=n_seg * SEG)
    assert len(got) == len(keep)
    for g, n in zip(sorted(got, key=lambda x: (x.program, x.pitch, x.onset)), sorted(keep, key=lambda x: (x.program, x.pitch, x.onset))):
        assert (g.is_drum, g.program, g.pitch) == (n.is_drum, n.program, n.pitch)
        assert abs(g.onset - n.onset) <= 0.0051 and (n.is_drum or abs(g.offset - n.offset) <= 0.0051)


def test_untied_note_is_closed_at_the_next_segment_start_and_dangling_offsets_are_dropped():
    tm = TaskManager()
    c = tm.codec
    seg0 = [c.encode(Event("tie", 0)), c.encode(Event("shift", 50)), c.encode(Event("velocity", 1)), c.encode(Event("program", 0)),
            c.encode(Event("pitch", 60)), EOS]
    seg1 = [c.encode(Event("tie", 0)), c.encode(Event("shift", 10)), c.encode(Event("velocity", 0)), c.encode(Event("pitch", 72)), EOS]
    segs = tm.detokenize_list_batches([np.array([seg0 + [PAD] * 4]), np.array([seg1 + [PAD] * 5])], [0.0, SEG])
    notes = note_events_to_notes(segs, end_sec=2 * SEG)
    assert notes == [Note(0.5, SEG, False, 0, 60)]


def test_malformed_tokens_are_counted_not_fatal():
    tm = TaskManager()
    c = tm.codec
    row = np.array([c.encode(Event("drum", 36)), UNK, 1500, c.encode(Event("tie", 0)), c.encode(Event("shift", 3)),
                    c.encode(Event("drum", 38)), EOS, c.encode(Event("pitch", 1))])
    segs, bad = tm.detokenize_list_batches([row[None]], [0.0], return_events=True)
    assert bad == 3                                          # drum inside the tie section, UNK, id beyond the codec
    assert segs[0][1] == [NoteEvent(0.03, True, DRUM_PROGRAM, 1, 38)]
    with pytest.raises(ValueError):
        tm.detokenize_list_batches([row[None]], [0.0, 1.0])


def test_multichannel_task_and_program_groups():
    tm = TaskManager("mc13_full_plus_256")
    assert tm.num_decoding_channels == 13 == len(MC13_GROUPS) and tm.max_note_token_length == 256
    assert tm.channel_of_program(0) == 0 and tm.channel_of_program(33) == 4 and tm.channel_of_program(128) == 12
    assert tm.channel_of_program(129) == 11
    with pytest.raises(ValueError):
        TaskManager("nope")
    c = tm.codec
    toks = np.zeros((1, 13, 16), dtype=np.int32)
    for ch, prog in ((0, 0), (4, 33)):
        toks[0, ch, :7] = [c.encode(Event("tie", 0)), c.encode(Event("shift", 10)), c.encode(Event("velocity", 1)),
                           c.encode(Event("program", prog)), c.encode(Event("pitch", 50 + ch)), c.encode(Event("shift", 10)), EOS]
    notes = tm.tokens_to_notes([toks], [0.0], end_sec=1.0)
    assert [(n.program, n.pitch, n.onset, n.offset) for n in notes] == [(0, 50, 0.1, 1.0), (33, 54, 0.1, 1.0)]


def test_midi_roundtrip(tmp_path):
    notes = [Note(0.1, 0.5, False, 0, 60), Note(0.3, 2.5, False, 40, 64), Note(1.0, 1.01, True, DRUM_PROGRAM, 36),
             Note(2.5, 3.0, False, 0, 60)]
    data = notes_to_midi_bytes(notes)
    assert data[:4] == b"MThd" and data.count(b"MTrk") == 4            # tempo + 3 programs
    back = read_midi_notes(data)
    assert len(back) == len(notes)
    for a, b in zip(back, sorted(notes)):
        assert (a.is_drum, a.program, a.pitch) == (b.is_drum, b.program, b.pitch)
        assert abs(a.onset - b.onset) < 2e-3 and abs(a.offset - b.offset) < 2e-3
    p = write_midi(notes, str(tmp_path / "x.mid"))
    assert open(p, "rb").read() == data


def _channel_programs(data):
    """{channel: set of programs} and {channel: note-on count} of a file written by notes_to_midi_bytes"""
    import struct
    progs, ons, pos = {}, {}, 14
    ntrk = struct.unpack(">H", data[10:12])[0]
    for _ in range(ntrk):
        ln = struct.unpack(">I", data[pos + 4:pos + 8])[0]
        p, end = pos + 8, pos + 8 + ln
        while p < end:
            while data[p] & 0x80:
                p += 1
            p += 1
            st = data[p]; p += 1
            if st == 0xFF:
                p += 2 + data[p + 1]
            elif st & 0xF0 == 0xC0:
                progs.setdefault(st & 15, set()).add(data[p]); p += 1
            else:
                if st & 0xF0 == 0x90:
                    ons[st & 15] = ons.get(st & 15, 0) + 1
                p += 2
        pos = end
    return progs, ons


def test_more_programs_than_midi_channels_merge_by_family_never_alias():
    """22 melodic programs + drums + singing: 15 melodic channels, exactly one program each; programs of one GM family share
    the family's lowest program; no note is lost; the singing voice (program 129) is written as GM 53."""
    from yourmt3_amd.midi import gm_program
    progs = [0, 1, 2, 3, 8, 9, 16, 17, 24, 25, 26, 32, 33, 40, 41, 48, 56, 64, 72, 80, 88, 96]
    notes = [Note(0.1 * i, 0.1 * i + 0.05, False, p, 40 + i) for i, p in enumerate(progs)]
    notes += [Note(0.2, 0.3, False, 129, 70), Note(0.5, 0.51, True, DRUM_PROGRAM, 36)]
    data = notes_to_midi_bytes(notes)
    by_ch, ons = _channel_programs(data)
    assert 9 not in by_ch and len(by_ch) <= 15 and all(len(v) == 1 for v in by_ch.values())
    assert sum(ons.values()) == len(notes) and ons[9] == 1
    written = {next(iter(v)) for v in by_ch.values()}
    assert gm_program(129) == 53 and 53 in written
    assert {0, 24} <= written and not ({1, 2, 3, 25, 26} & written)           # the largest families were folded first
    assert notes_to_midi_bytes(notes) == data                                   # deterministic
    few = [Note(0.0, 0.1, False, p, 60) for p in range(15)] + [Note(0.0, 0.1, False, 129, 61)]
    by_ch, _ = _channel_programs(notes_to_midi_bytes(few[:15]))
    assert sorted(next(iter(v)) for v in by_ch.values()) == list(range(15))   # 15 programs: untouched, one channel each
    with pytest.raises(ValueError):
        gm_program(130)


def test_slice_padded_array_and_wav_ingest(tmp_path):
    x = np.arange(70000, dtype=np.float32)
    s = slice_padded_array(x, 32767)
    assert s.shape == (3, 1, 32767) and s[2, 0, 70000 - 2 * 32767 - 1] == 69999 and s[2, 0, 70000 - 2 * 32767] == 0
    assert slice_padded_array(np.zeros(0, np.float32), 32767).shape == (1, 1, 32767)
    assert slice_padded_array(np.zeros(32767, np.float32), 32767).shape == (1, 1, 32767)
    path = str(tmp_path / "t.wav")
    t = np.arange(44100) / 44100.0
    stereo = np.stack([np.sin(2 * np.pi * 440 * t), np.sin(2 * np.pi * 440 * t)], 1)
    with wave.open(path, "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(44100)
        w.writeframes((stereo * 32767).astype("<i2").tobytes())
    y, sr = load_wav(path)
    assert sr == 44100 and y.shape == (44100,) and abs(np.abs(y).max() - 1.0) < 1e-3
    z = resample(y, sr, 16000)
    assert z.shape == (16000,) and z.dtype == np.float32
    k = np.fft.rfft(z[:16000]).__abs__().argmax()
    assert k == 440


class _FakeModel:
    """Stands in for YourMT3 on the CPU: returns the token arrays a perfect model would emit."""
    def __init__(self, cfg, rows):
        self.cfg, self.rows, self.calls = cfg, rows, []

    def ingest(self, pcm, sample_rate):
        x = resample(pcm.numpy().reshape(pcm.shape[0], -1).mean(axis=1), sample_rate, self.cfg.sample_rate)
        self.last_ingest_samples = x.shape[0]
        return slice_padded_array(x, self.cfg.segment_samples)

    def inference_file(self, bsz, segments, max_token_length=None):
        self.calls.append((bsz, tuple(segments.shape), max_token_length))
        return [self.rows[i:i + bsz] for i in range(0, self.rows.shape[0], bsz)]


def test_transcribe_flow_writes_the_expected_midi(tmp_path):
    cfg = YMT3Config()
    tm = TaskManager()
    notes = [Note(0.5, 1.0, False, 0, 60), Note(1.5, 3.0, False, 0, 64), Note(2.5, 2.51, True, DRUM_PROGRAM, 36)]
    toks = _tokenize(tm, notes, 2, 1024)[:, None, :]
    model = _FakeModel(cfg, toks)
    audio = np.zeros(int(3.5 * 16000), dtype=np.float32)
    path, got = transcribe(model, audio, task_manager=tm, bsz=1, output_dir=str(tmp_path), return_notes=True)
    assert model.calls == [(1, (2, 1, 32767), 1024)]
    assert os.path.basename(path) == "audio.mid" and len(got) == 3
    back = read_midi_notes(open(path, "rb").read())
    assert [(n.program, n.pitch) for n in back] == [(0, 60), (0, 64), (DRUM_PROGRAM, 36)]
    assert abs(back[1].onset - 1.5) < 6e-3 and abs(back[1].offset - 3.0) < 6e-3
    with pytest.raises(ValueError):
        transcribe(model, audio, task_manager=TaskManager("mc13_full_plus_256"))


# ------------------------------------------------------------------------------------------------ property tests
from hypothesis import given, settings, strategies as st  # noqa: E402


@settings(max_examples=200, deadline=None)
@given(st.lists(st.lists(st.integers(min_value=-5, max_value=1600), min_size=0, max_size=96), min_size=1, max_size=4))
def test_detokenizer_accepts_any_id_stream(segments):
    """Whatever ids the model emits -- out of vocabulary, negative, offsets without onsets, programs without pitches,
    shifts past the segment -- detokenising never raises, reports malformed tokens as a count, and yields well-formed notes."""
    tm = TaskManager()
    L = max(1, max(len(s) for s in segments))
    arr = np.full((len(segments), 1, L), PAD, dtype=np.int32)
    for i, sgm in enumerate(segments):
        arr[i, 0, :len(sgm)] = sgm
    starts = [i * SEG for i in range(len(segments))]
    for i, sgm in enumerate(segments):
        ev, ties, bad = tm.tokenizer.decode_segment(list(arr[i, 0]), starts[i])
        assert bad >= 0 and all(isinstance(e, NoteEvent) for e in ev)
        assert all(e.time >= starts[i] - 1e-9 for e in ev)
    notes = tm.tokens_to_notes([arr], starts, end_sec=len(segments) * SEG)
    for n in notes:
        assert 0 <= n.pitch < 128 and n.offset >= n.onset >= 0.0
        assert n.is_drum == (n.program == DRUM_PROGRAM)
    data = notes_to_midi_bytes(notes)
    back = read_midi_notes(data)                      # same-pitch notes that overlap or coincide are merged by the writer
    assert data[:4] == b"MThd" and len(back) <= len(notes) and (len(back) > 0) == (len(notes) > 0)
    assert all(n.offset > n.onset for n in back)


_note = st.tuples(st.integers(0, 740), st.integers(5, 300), st.sampled_from([0, 24, 40, 129]), st.integers(21, 108), st.booleans())


@settings(max_examples=60, deadline=None)
@given(st.lists(_note, min_size=1, max_size=40))
def test_tokenise_detokenise_roundtrip_property(raw):
    """Random note sets over 4 segments (10 ms grid, ties across segment boundaries, drums): tokenise per segment ->
    detokenise -> the same notes, onsets and offsets within half a grid step."""
    tm = TaskManager()
    notes, busy = [], {}
    for on_cs, dur_cs, prog, pitch, drum in sorted(raw):
        on = on_cs / 100.0
        if drum:
            notes.append(Note(on, on + 0.01, True, DRUM_PROGRAM, 35 + pitch % 10))
            continue
        off = min(on + dur_cs / 100.0, 4 * SEG - 0.05)
        k = (prog, pitch)
        if off <= on + 0.02 or (k in busy and busy[k] >= on - 0.02):
            continue                                  # overlapping notes of one (program, pitch) cannot round-trip exactly
        busy[k] = off
        notes.append(Note(on, off, False, prog, pitch))
    if not notes:
        return
    toks = _tokenize(tm, notes, 4, 1024)
    got = tm.tokens_to_notes([toks[:, None, :]], [i * SEG for i in range(4)], end_sec=4 * SEG)
    key = lambda x: (x.is_drum, x.program, x.pitch, round(x.onset, 2))
    # drum hits on the same pitch and grid step collapse to one event in the token stream: compare sets of keys
    assert {key(n) for n in got} == {key(Note(_grid(n.onset, SEG * int(n.onset // SEG)), n.offset, n.is_drum, n.program, n.pitch)) for n in notes}
    for g in got:
        if not g.is_drum:
            m = [n for n in notes if (n.program, n.pitch) == (g.program, g.pitch) and abs(n.onset - g.onset) <= 0.0051]
            assert m and abs(m[0].offset - g.offset) <= 0.0051
