"""Host-side tables of the log-mel frontend (window + mel filterbank).

torchaudio is not a dependency; these are the buffers
``torchaudio.transforms.MelSpectrogram(sample_rate, n_fft, hop_length, f_min,
n_mels)`` registers (``spectrogram.window`` and ``mel_scale.fb``; the call site
is ref: music2midi/input.py:25-31), rebuilt from the published formulas in
float32: periodic Hann window; HTK mel scale, ``norm=None``,
``f_max = sample_rate // 2``.  The two roundings that decide the last bit of a
filter weight (``linspace`` and ``10 ** x`` in float32) are done with the same
torch CPU ops torchaudio itself uses, so the table equals torchaudio's for the
installed torch; a 1-ulp change of a band edge near 8 kHz moves a weight by
~2e-5, which would eat into the 1e-4 log-mel tolerance.  A real checkpoint carries both buffers in its
state dict and ``LogMelSpectrogram.load_buffers`` takes them
verbatim instead.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def hann_window(n_fft: int) -> np.ndarray:
    """The buffer torchaudio's Spectrogram registers: ``torch.hann_window(n_fft)`` (periodic, float32).

    It is evaluated BY torch in float32, which rounds 1 356 of the 2 048 taps differently (up to 2e-7) from the
    correctly rounded float64 formula.  That matters: on a tonal frame a 1e-7 window perturbation leaks ~1e-7 of
    the peak into every bin, i.e. 2e-4 in the log domain on a bin 45 dB below the peak (found by the music-like
    parity fixture) — so the table must be torch's own, bit for bit, as a reference checkpoint carries it."""
    return torch.hann_window(n_fft, periodic=True, dtype=torch.float32).numpy().copy()


def _linspace_f32(start: float, end: float, steps: int) -> np.ndarray:
    return torch.linspace(start, end, steps, dtype=torch.float32).numpy()


def mel_filterbank(sample_rate: int, n_fft: int, f_min: float, n_mels: int) -> np.ndarray:
    """[n_fft//2+1, n_mels] float32 triangular HTK filterbank (norm=None)."""
    n_freqs = n_fft // 2 + 1
    f_max = float(sample_rate // 2)
    all_freqs = _linspace_f32(0.0, float(sample_rate // 2), n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = _linspace_f32(m_min, m_max, n_mels + 2)
    f_pts = (700.0 * (10.0 ** (torch.from_numpy(m_pts) / 2595.0) - 1.0)).numpy()
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = (-slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(np.float32(0.0), np.minimum(down, up)).astype(np.float32)
