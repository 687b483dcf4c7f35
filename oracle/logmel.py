"""Oracle: log-mel frontend (test infrastructure, see oracle/__init__.py).

Follows ref: music2midi/input.py:15-41, whose arithmetic is
``torchaudio.transforms.MelSpectrogram`` (torchaudio 2.1.0, not in the image).
Restated from that package's published source:

* ``Spectrogram``: ``torch.stft(x, n_fft, hop, win_length=n_fft,
  window=hann_window(n_fft) (periodic), center=True, pad_mode="reflect",
  normalized=False, onesided=True, return_complex=True)`` then ``|.|**2``
  (power=2.0).
* ``MelScale``: ``fb = melscale_fbanks(n_fft//2+1, f_min, f_max=sample_rate//2,
  n_mels, sample_rate, norm=None, mel_scale="htk")`` and
  ``mel = (spec^T @ fb)^T``.
* ref input.py:39-40: transpose to [B, frames, n_mels], ``clamp(min=1e-6)``,
  ``log``.
"""
from __future__ import annotations

import math

import torch


def melscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int,
                    sample_rate: int) -> torch.Tensor:
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') -> [n_freqs, n_mels]."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + (f_min / 700.0))
    m_max = 2595.0 * math.log10(1.0 + (f_max / 700.0))
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    zero = torch.zeros(1)
    down_slopes = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up_slopes = slopes[:, 2:] / f_diff[1:]
    return torch.max(zero, torch.min(down_slopes, up_slopes))


class LogMelOracle:
    def __init__(self, sample_rate: int, n_fft: int, hop_length: int, f_min: float, n_mels: int):
        self.n_fft = n_fft
        self.hop = hop_length
        self.window = torch.hann_window(n_fft)
        self.fb = melscale_fbanks(n_fft // 2 + 1, f_min, float(sample_rate // 2), n_mels, sample_rate)

    def power_spectrogram(self, x: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
        x = x.to(dtype)
        spec = torch.stft(x, self.n_fft, hop_length=self.hop, win_length=self.n_fft,
                          window=self.window.to(dtype), center=True, pad_mode="reflect",
                          normalized=False, onesided=True, return_complex=True)
        return spec.abs().pow(2.0)  # [B, n_freqs, frames]

    def __call__(self, x: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
        """x [B, T] -> log-mel [B, frames, n_mels] (dtype float32, or float64 as a truth probe)."""
        with torch.no_grad():
            p = self.power_spectrogram(x, dtype)
            mel = torch.matmul(p.transpose(-1, -2), self.fb.to(dtype))  # [B, frames, n_mels]
            return mel.clamp(min=1e-6).log()


def conditioning(feature: torch.Tensor, indices: torch.Tensor, embeds) -> torch.Tensor:
    """ref: music2midi/input.py:50-59 — cond tokens FIRST, then the feature rows."""
    rows = [e[indices[:, i]] for i, e in enumerate(embeds)]
    return torch.cat([torch.stack(rows, dim=1), feature], dim=1)
