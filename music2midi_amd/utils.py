"""Note array -> MIDI object (ref: music2midi/utils.py:5-20).

With ``pretty_midi`` installed this returns a real ``PrettyMIDI`` exactly as the
reference does (resolution 384, 120 bpm, one piano instrument, invalid notes
removed).  The image used for the MI355X build has no pretty_midi, so a small
stand-in with the members the reference's callers touch (``instruments[0].notes``,
``get_end_time``, ``write``) is returned instead; ``fluidsynth`` needs the real
package and says so.
"""
from __future__ import annotations

import struct
from typing import List

import numpy as np

try:  # optional
    import pretty_midi as _pm
except Exception:  # pragma: no cover - depends on the image
    _pm = None


class SimpleNote:
    __slots__ = ("start", "end", "pitch", "velocity")

    def __init__(self, start, end, pitch, velocity):
        self.start, self.end, self.pitch, self.velocity = float(start), float(end), int(pitch), int(velocity)


class SimpleInstrument:
    def __init__(self, program=0, name="Piano"):
        self.program, self.name, self.is_drum = program, name, False
        self.notes: List[SimpleNote] = []


class SimpleMIDI:
    """Just enough of PrettyMIDI for the callers of Music2MIDI.generate."""

    def __init__(self, resolution=384, initial_tempo=120.0):
        self.resolution, self.initial_tempo = resolution, initial_tempo
        self.instruments: List[SimpleInstrument] = []

    def remove_invalid_notes(self):
        for inst in self.instruments:
            inst.notes = [n for n in inst.notes if n.end > n.start]

    def get_end_time(self) -> float:
        ends = [n.end for inst in self.instruments for n in inst.notes]
        return max(ends) if ends else 0.0

    def note_array(self) -> np.ndarray:
        rows = [[n.start, n.end, n.pitch, n.velocity] for inst in self.instruments for n in inst.notes]
        return np.asarray(rows, dtype=np.float64).reshape(-1, 4)

    def write(self, path) -> None:
        """Standard MIDI file, format 0, one track."""
        ticks_per_second = self.resolution * self.initial_tempo / 60.0
        events = []
        for inst in self.instruments:
            for n in inst.notes:
                events.append((int(round(n.start * ticks_per_second)), 1, 0x90, n.pitch, max(1, min(127, n.velocity))))
                events.append((int(round(n.end * ticks_per_second)), 0, 0x80, n.pitch, 0))
        events.sort(key=lambda e: (e[0], e[1]))

        def vlq(v):
            out = [v & 0x7F]
            v >>= 7
            while v:
                out.append((v & 0x7F) | 0x80)
                v >>= 7
            return bytes(reversed(out))

        tempo = int(round(60_000_000 / self.initial_tempo))
        track = b"\x00\xff\x51\x03" + struct.pack(">I", tempo)[1:]
        track += b"\x00\xc0" + bytes([self.instruments[0].program if self.instruments else 0])
        last = 0
        for tick, _, status, pitch, vel in events:
            track += vlq(tick - last) + bytes([status, pitch & 0x7F, vel & 0x7F])
            last = tick
        track += b"\x00\xff\x2f\x00"
        with open(path, "wb") as f:
            f.write(b"MThd" + struct.pack(">IHHH", 6, 0, 1, self.resolution))
            f.write(b"MTrk" + struct.pack(">I", len(track)) + track)

    def fluidsynth(self, *a, **k):
        raise ImportError("audio rendering needs the real pretty_midi + fluidsynth packages")


def numpy_to_midi(notes: np.ndarray):
    """rows (onset_s, offset_s, pitch, velocity) -> PrettyMIDI (or the stand-in above)."""
    if _pm is not None:
        midi = _pm.PrettyMIDI(resolution=384, initial_tempo=120.0)
        inst = _pm.Instrument(program=0, name="Piano")
        inst.notes = [_pm.Note(start=s, end=e, pitch=int(p), velocity=int(v)) for s, e, p, v in notes]
    else:
        midi = SimpleMIDI(resolution=384, initial_tempo=120.0)
        inst = SimpleInstrument(program=0, name="Piano")
        inst.notes = [SimpleNote(s, e, p, v) for s, e, p, v in notes]
    midi.instruments.append(inst)
    midi.remove_invalid_notes()
    return midi
