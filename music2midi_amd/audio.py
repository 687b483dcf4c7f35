"""Audio ingest and the dataset-side augmentation, without librosa / soundfile.

* ``load_audio(path, sr)`` — what ``librosa.load(path, sr=sr)`` gives the reference
  (ref: music2midi/model.py:83-84, dataset.py:124-129): mono float32 at ``sr``.  librosa is used when
  it is installed (bit-for-bit the reference's ingest, any container it can decode); otherwise RIFF/WAVE
  files are parsed here — PCM 8/16/24/32-bit, IEEE float 32/64-bit, ``WAVE_FORMAT_EXTENSIBLE``,
  any channel count — and resampled with a polyphase filter.  Compressed containers (mp3/mp4, which
  ``webui.py`` feeds through librosa's audioread backend) need librosa/ffmpeg and fail loudly without.
* ``pitch_shift`` / ``transpose`` / ``normalize`` — ref: music2midi/dataset.py:131-133,157-160
  (``librosa.effects.pitch_shift``: phase-vocoder time stretch by 2^(-n/12), resample back, fix length;
  ``librosa.util.normalize``: divide by max |y|).  Restated from librosa's published algorithm
  (n_fft 2048, hop 512, Hann, zero-padded centred frames); the resampler is scipy's polyphase filter
  rather than soxr, so samples are NOT bit-equal to librosa's — **parity unpinned** (librosa is absent
  from the build image); tests pin the properties (pitch ratio, length, energy).

Host-side numpy: this is the step BEFORE the hot path; the reference runs it in
DataLoader workers on the CPU as well.
"""
from __future__ import annotations

import struct
from math import gcd
from pathlib import Path
from typing import Tuple

import numpy as np

_WAVE_FORMAT_PCM = 1
_WAVE_FORMAT_IEEE_FLOAT = 3
_WAVE_FORMAT_EXTENSIBLE = 0xFFFE


def read_wav(path) -> Tuple[np.ndarray, int]:
    """RIFF/WAVE -> (float32 [n_frames, n_channels] in [-1, 1), sample rate)."""
    raw = Path(path).read_bytes()
    if len(raw) < 12 or raw[:4] != b"RIFF" or raw[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file (compressed formats need librosa/ffmpeg, which are not installed)")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(raw):
        cid, size = raw[pos:pos + 4], struct.unpack("<I", raw[pos + 4:pos + 8])[0]
        body = raw[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = body
        elif cid == b"data":
            data = body
        pos += 8 + size + (size & 1)              # chunks are word aligned
    if fmt is None or data is None or len(fmt) < 16:
        raise ValueError(f"{path}: missing 'fmt ' or 'data' chunk")
    tag, n_ch, rate, _, block_align, bits = struct.unpack("<HHIIHH", fmt[:16])
    if tag == _WAVE_FORMAT_EXTENSIBLE and len(fmt) >= 26:
        tag = struct.unpack("<H", fmt[24:26])[0]  # first two bytes of the sub-format GUID
    if n_ch < 1 or rate < 1:
        raise ValueError(f"{path}: bad channel count / sample rate")
    width = bits // 8
    n = len(data) // (width * n_ch) * n_ch
    if tag == _WAVE_FORMAT_PCM:
        if width == 1:
            y = (np.frombuffer(data, np.uint8, n).astype(np.float32) - 128.0) / 128.0
        elif width == 2:
            y = np.frombuffer(data, "<i2", n).astype(np.float32) / 32768.0
        elif width == 3:
            b = np.frombuffer(data, np.uint8, n * 3).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v & 0x800000, v - 0x1000000, v)
            y = v.astype(np.float32) / 8388608.0
        elif width == 4:
            y = (np.frombuffer(data, "<i4", n).astype(np.float64) / 2147483648.0).astype(np.float32)
        else:
            raise ValueError(f"{path}: unsupported PCM sample width {bits} bits")
    elif tag == _WAVE_FORMAT_IEEE_FLOAT:
        if width == 4:
            y = np.frombuffer(data, "<f4", n).astype(np.float32)
        elif width == 8:
            y = np.frombuffer(data, "<f8", n).astype(np.float32)
        else:
            raise ValueError(f"{path}: unsupported float sample width {bits} bits")
    else:
        raise ValueError(f"{path}: unsupported WAVE format tag {tag} (only PCM and IEEE float)")
    return y.reshape(-1, n_ch), int(rate)


def resample(y: np.ndarray, orig_sr: float, target_sr: float) -> np.ndarray:
    """Polyphase resampling along the last axis (rational approximation of the rate ratio)."""
    if orig_sr == target_sr:
        return y.astype(np.float32, copy=False)
    from fractions import Fraction
    from scipy.signal import resample_poly
    frac = Fraction(float(target_sr) / float(orig_sr)).limit_denominator(1000)
    up, down = frac.numerator, frac.denominator
    g = gcd(up, down)
    return resample_poly(y, up // g, down // g, axis=-1).astype(np.float32)


def load_audio(path, sr: int) -> np.ndarray:
    """Mono float32 at ``sr`` — the reference's ``librosa.load(str(path), sr=sr)[0]``."""
    try:
        import librosa  # type: ignore
        y, _ = librosa.load(str(path), sr=sr)
        return y
    except ImportError:
        pass
    y, rate = read_wav(path)
    y = y.mean(axis=1)                            # librosa.to_mono
    return resample(y, rate, sr).astype(np.float32)


def normalize(y: np.ndarray) -> np.ndarray:
    """librosa.util.normalize(y) with its defaults: divide by max |y| (left alone when that is ~0)."""
    peak = np.abs(y).max() if y.size else 0.0
    return y if peak < np.finfo(np.float32).tiny else (y / peak).astype(y.dtype)


# ------------------------------------------------------------------ phase vocoder
def _stft(y: np.ndarray, n_fft: int, hop: int) -> np.ndarray:
    pad = n_fft // 2
    yp = np.concatenate([np.zeros(pad, y.dtype), y, np.zeros(pad, y.dtype)])
    n_frames = 1 + (len(yp) - n_fft) // hop
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)       # periodic Hann
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    return np.fft.rfft(yp[idx] * win, axis=1).T                            # [1 + n_fft/2, n_frames]


def _istft(D: np.ndarray, n_fft: int, hop: int, length: int) -> np.ndarray:
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)
    frames = np.fft.irfft(D.T, n=n_fft, axis=1) * win
    n_frames = frames.shape[0]
    out = np.zeros(n_fft + hop * (n_frames - 1))
    norm = np.zeros_like(out)
    for i in range(n_frames):
        out[i * hop:i * hop + n_fft] += frames[i]
        norm[i * hop:i * hop + n_fft] += win * win
    out = out / np.where(norm > 1e-10, norm, 1.0)
    out = out[n_fft // 2:]
    if len(out) < length:
        out = np.concatenate([out, np.zeros(length - len(out))])
    return out[:length]


def _phase_vocoder(D: np.ndarray, rate: float, hop: int, n_fft: int) -> np.ndarray:
    """librosa.phase_vocoder: magnitudes interpolated linearly between frames, phases advanced by the
    unwrapped frame-to-frame increment."""
    n_bins, n_frames = D.shape
    steps = np.arange(0, n_frames, rate, dtype=np.float64)
    out = np.zeros((n_bins, len(steps)), dtype=np.complex128)
    omega = np.linspace(0, np.pi * hop, n_bins)                            # expected phase advance per hop
    acc = np.angle(D[:, 0])
    Dp = np.concatenate([D, np.zeros((n_bins, 2), dtype=D.dtype)], axis=1)
    for t, step in enumerate(steps):
        k = int(step)
        a, b = Dp[:, k], Dp[:, k + 1]
        alpha = step - k
        mag = (1.0 - alpha) * np.abs(a) + alpha * np.abs(b)
        out[:, t] = mag * np.exp(1j * acc)
        dphase = np.angle(b) - np.angle(a) - omega
        dphase = dphase - 2.0 * np.pi * np.round(dphase / (2.0 * np.pi))
        acc = acc + omega + dphase
    return out


def time_stretch(y: np.ndarray, rate: float, n_fft: int = 2048, hop: int = 512) -> np.ndarray:
    """librosa.effects.time_stretch: rate > 1 shortens."""
    if rate <= 0:
        raise ValueError("rate must be a positive number")
    D = _stft(np.asarray(y, dtype=np.float64), n_fft, hop)
    return _istft(_phase_vocoder(D, rate, hop, n_fft), n_fft, hop, int(round(len(y) / rate))).astype(np.float32)


def pitch_shift(y: np.ndarray, sr: float, n_steps: float, bins_per_octave: int = 12) -> np.ndarray:
    """librosa.effects.pitch_shift(y, sr=sr, n_steps=n_steps): same length, pitch moved by n_steps semitones."""
    y = np.asarray(y, dtype=np.float32)
    if n_steps == 0:
        return y.copy()
    rate = 2.0 ** (-float(n_steps) / bins_per_octave)
    shifted = resample(time_stretch(y, rate), float(sr) / rate, sr)
    out = np.zeros(len(y), dtype=np.float32)                               # librosa.util.fix_length
    n = min(len(y), len(shifted))
    out[:n] = shifted[:n]
    return out


def transpose(waveform: np.ndarray, notes: np.ndarray, step: int, sr: float):
    """ref: music2midi/dataset.py:157-160 — shift the audio AND the note pitches by ``step`` semitones."""
    waveform = pitch_shift(waveform, sr=sr, n_steps=step)
    notes = np.array(notes, dtype=np.float64, copy=True)
    notes[:, 2] += step
    return waveform, notes
