"""Oracle: melody chroma accuracy (test infrastructure, see oracle/__init__.py).

Follows ref: music2midi/evaluation.py:10-75, whose arithmetic lives in third-party packages that are
not under /root/reference and not in the image (ref: environment.yaml:228,417):

* ``pretty_midi==0.2.10``  ``Instrument.get_piano_roll(fs, times)`` — a [128, int(fs*end)] roll with
  ``roll[pitch, int(start*fs):int(end*fs)] += velocity`` per note, then re-sampled at ``times``:
  column n = mean of the roll columns ``[round(times[n]*fs), round(times[n+1]*fs))`` (at least one),
  columns at or past the roll's end stay zero, and so does the LAST column (the loop pairs
  ``times[:-1]`` with ``times[1:]``).
* ``mir_eval==0.6``  ``melody.hz2cents`` (1200*log2(f/10 Hz), 0 stays 0), ``freq_to_voicing``
  (voiced = f > 0), ``to_cent_voicing`` on identical time bases (no resampling), and
  ``raw_chroma_accuracy`` (octave-folded cent difference < 50 on frames where both have a frequency,
  counted over the voiced reference frames).
* ``librosa.midi_to_hz``  440 * 2^((p-69)/12).

Written as plain per-frame loops on purpose — an independent restatement to check the vectorised
product code against.  **Parity unpinned**: neither package can be imported here, and the reference
holds no test vectors for this path.

Silent frames: ref evaluation.py:15-18 stores NaN into an int array and then indexes the last element
of an EMPTY array under ``numba.njit`` (no bounds check): undefined values.  This oracle (and the
product) read such a frame as *unvoiced* (frequency 0), the reading the metric's own voicing logic
expects; the difference is documented in DESIGN_HISTORY.md (section 7), not imitated.
"""
from __future__ import annotations

import math

import numpy as np


def instrument_piano_roll(notes, fs, times):
    """notes: rows (start, end, pitch, velocity) -> [128, len(times)]."""
    notes = [tuple(n) for n in np.asarray(notes, dtype=np.float64).reshape(-1, 4) if n[1] > n[0]]
    out = np.zeros((128, len(times)))
    if not notes:
        return out
    end_time = max(n[1] for n in notes)
    roll = np.zeros((128, int(fs * end_time)))
    for start, end, pitch, vel in notes:
        roll[int(pitch), int(start * fs):int(end * fs)] += int(vel)
    t = np.array(np.round(np.asarray(times) * fs), dtype=np.int64)
    for n in range(len(t) - 1):
        start, end = int(t[n]), int(t[n + 1])
        if start < roll.shape[1]:
            if start == end:
                end = start + 1
            out[:, n] = roll[:, start:end].mean(axis=1)
    return out


def highest_pitch_per_frame(roll):
    out = []
    for i in range(roll.shape[1]):
        on = np.nonzero(roll[:, i])[0]
        out.append(int(on[-1]) if len(on) else -1)      # -1: silent frame, read as unvoiced
    return np.asarray(out, dtype=np.int64)


def extract_midi_melody(target_notes, output_notes, fs=100):
    def end(n):
        n = np.asarray(n, dtype=np.float64).reshape(-1, 4)
        n = n[n[:, 1] > n[:, 0]]
        return float(n[:, 1].max()) if len(n) else 0.0
    times = np.arange(0, max(end(target_notes), end(output_notes)), 1 / fs)
    return (highest_pitch_per_frame(instrument_piano_roll(target_notes, fs, times)),
            highest_pitch_per_frame(instrument_piano_roll(output_notes, fs, times)))


def _hz(p):
    return 0.0 if p < 0 else 440.0 * 2.0 ** ((p - 69) / 12.0)


def raw_chroma_accuracy(ref_pitch, est_pitch, cent_tolerance=50.0):
    assert len(ref_pitch) == len(est_pitch)
    ref_hz = [_hz(p) for p in ref_pitch]
    est_hz = [_hz(p) for p in est_pitch]
    ref_voiced = [1.0 if f > 0 else 0.0 for f in ref_hz]
    ref_cent = [1200.0 * math.log2(f / 10.0) if f > 0 else 0.0 for f in ref_hz]
    est_cent = [1200.0 * math.log2(f / 10.0) if f > 0 else 0.0 for f in est_hz]
    if len(ref_voiced) == 0 or sum(ref_voiced) == 0:
        return 0.0
    hits, both = 0.0, 0
    for rv, rc, ec in zip(ref_voiced, ref_cent, est_cent):
        if rc == 0 or ec == 0:
            continue
        both += 1
        diff = abs(rc - ec)
        octave = 1200.0 * math.floor(diff / 1200.0 + 0.5)
        if abs(diff - octave) < cent_tolerance:
            hits += rv
    if both == 0:
        return 0.0
    return hits / sum(ref_voiced)


def evaluate_batch(target_note_sets, output_note_sets):
    ts, os_ = [], []
    for t, o in zip(target_note_sets, output_note_sets):
        a, b = extract_midi_melody(t, o)
        ts.append(a)
        os_.append(b)
    if not ts:
        return 0.0
    return raw_chroma_accuracy(np.concatenate(ts), np.concatenate(os_))
