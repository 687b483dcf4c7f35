"""Melody chroma accuracy (ref: music2midi/evaluation.py:10-75) without
librosa / mir_eval / numba / pretty_midi.

Pipeline restated: piano roll at 100 frames/s -> highest sounding pitch per
frame -> ``mir_eval.melody.raw_chroma_accuracy`` between target and output.

One deliberate difference, documented rather than imitated:
for a frame in which nothing sounds the reference indexes ``onset_pitches[-1]``
of an EMPTY array inside ``numba.njit`` (no bounds check) after storing NaN into
an int array (ref evaluation.py:15-18) — undefined values.  Here such a frame is
*unvoiced* (frequency 0), which is what the metric's voicing logic expects.
"""
from __future__ import annotations

from typing import Iterable, Tuple

import numpy as np


def _notes_of(midi) -> np.ndarray:
    if isinstance(midi, np.ndarray):
        return midi.reshape(-1, 4)
    rows = [[n.start, n.end, n.pitch, n.velocity] for inst in midi.instruments for n in inst.notes]
    return np.asarray(rows, dtype=np.float64).reshape(-1, 4)


def _end_time(notes: np.ndarray) -> float:
    return float(notes[:, 1].max()) if len(notes) else 0.0


def piano_roll(notes: np.ndarray, n_frames: int, fs: int = 100) -> np.ndarray:
    """[128, n_frames] velocity sums; a note sounds in frames [int(start*fs), int(end*fs)).
    The last frame stays zero, as pretty_midi's ``get_piano_roll(times=...)`` leaves it."""
    roll = np.zeros((128, n_frames))
    for start, end, pitch, vel in notes:
        a, b = int(start * fs), int(end * fs)
        roll[int(pitch), a:min(b, n_frames - 1)] += vel
    return roll


def highest_pitches(roll: np.ndarray) -> np.ndarray:
    """Highest sounding pitch per frame, -1 where the frame is silent."""
    active = roll > 0
    idx = 127 - np.argmax(active[::-1], axis=0)
    return np.where(active.any(axis=0), idx, -1).astype(np.int64)


def extract_midi_melody(target, output, fs: int = 100) -> Tuple[np.ndarray, np.ndarray]:
    tn, on = _notes_of(target), _notes_of(output)
    end_time = max(_end_time(tn), _end_time(on))
    n_frames = len(np.arange(0, end_time, 1 / fs))
    return highest_pitches(piano_roll(tn, n_frames, fs)), highest_pitches(piano_roll(on, n_frames, fs))


def _cents(pitch: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    voiced = pitch >= 0
    hz = np.where(voiced, 440.0 * 2.0 ** ((pitch - 69) / 12.0), 0.0)
    cent = np.zeros_like(hz)
    cent[voiced] = 1200.0 * np.log2(hz[voiced] / 10.0)   # mir_eval.melody.hz2cents base 10 Hz
    return voiced, cent


def melody_chroma_accuracy(ref_pitch: np.ndarray, est_pitch: np.ndarray, cent_tolerance: float = 50.0) -> float:
    assert ref_pitch.shape == est_pitch.shape
    ref_v, ref_c = _cents(ref_pitch)
    est_v, est_c = _cents(est_pitch)
    if ref_v.size == 0 or ref_v.sum() == 0:
        return 0.0
    nonzero = np.logical_and(est_c != 0, ref_c != 0)
    if nonzero.sum() == 0:
        return 0.0
    diff = np.abs(ref_c - est_c)[nonzero]
    octave = 1200.0 * np.floor(diff / 1200.0 + 0.5)
    correct = np.abs(diff - octave) < cent_tolerance
    return float(np.sum(ref_v[nonzero] * correct) / np.sum(ref_v))


def evaluate_batch(targets: Iterable, outputs: Iterable) -> float:
    pairs = [extract_midi_melody(t, o) for t, o in zip(targets, outputs)]
    if not pairs:
        return 0.0
    ts, os_ = zip(*pairs)
    return melody_chroma_accuracy(np.concatenate(ts), np.concatenate(os_))
