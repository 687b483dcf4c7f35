"""Model inputs, log-mel frontend and conditioning — MI355X implementations of
ref: music2midi/input.py (same class names, constructor arguments and tensor
contracts).  The arithmetic runs in hand-written HIP kernels behind the C ABI
(csrc/frontend.hip); torch is used for device buffers and streams only.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional

import numpy as np
import torch
import torch.nn as nn

from . import melbank, native


class ModelInputs(NamedTuple):
    """ref: music2midi/input.py:9-12."""
    input_waveform: torch.Tensor
    notes_batch: Optional[tuple] = None
    cond_index: Optional[torch.Tensor] = None


class LogMelSpectrogram(nn.Module):
    """waveform [B, T] -> log-mel [B, frames, n_mels] (ref: music2midi/input.py:15-41).

    Buffer names mirror torchaudio's so a reference checkpoint's
    ``spectrogram.melspectrogram.spectrogram.window`` / ``...mel_scale.fb``
    load into this module unchanged.
    """

    def __init__(self, sample_rate: int, n_fft: int, hop_length: int, f_min: float, n_mels: int):
        super().__init__()
        self.sample_rate = int(sample_rate)
        self.n_fft = int(n_fft)
        self.hop_length = int(hop_length)
        self.f_min = float(f_min)
        self.n_mels = int(n_mels)
        self.melspectrogram = _MelSpectrogramBuffers(self.sample_rate, self.n_fft, self.f_min, self.n_mels)
        self._plan = None           # native handle, built lazily from the current buffers
        self._plan_key = None

    # -- native plan ---------------------------------------------------------
    def _get_plan(self):
        win = self.melspectrogram.spectrogram.window
        fb = self.melspectrogram.mel_scale.fb
        key = (win.data_ptr(), fb.data_ptr(), win._version, fb._version)
        if self._plan is None or self._plan_key != key:
            self._free_plan()
            lib = native.load()
            w = np.ascontiguousarray(win.detach().cpu().numpy(), dtype=np.float32)
            f = np.ascontiguousarray(fb.detach().cpu().numpy(), dtype=np.float32)
            desc = native.FrontendDesc(self.n_fft, self.hop_length, self.n_fft // 2 + 1, self.n_mels,
                                       w.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p))
            h = C.c_void_p()
            native.check(lib.m2m_frontend_create(C.byref(desc), C.byref(h)), "m2m_frontend_create")
            self._plan, self._plan_key = h, key
        return self._plan

    def _free_plan(self):
        if self._plan is not None:
            native.load().m2m_frontend_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self._free_plan()
        except Exception:
            pass

    def num_frames(self, n_samples: int) -> int:
        return 1 + n_samples // self.hop_length

    # -- forward -------------------------------------------------------------
    def forward_into(self, x: torch.Tensor, out: torch.Tensor, row_offset: int = 0) -> torch.Tensor:
        """Write log-mel rows into ``out[:, row_offset:row_offset+frames, :]`` (no concat copy)."""
        native.require_gpu()
        if x.dim() != 2:
            raise ValueError(f"waveform must be (batch, sample), got shape {tuple(x.shape)}")
        if not x.is_cuda:
            raise native.NativeError("LogMelSpectrogram runs on the GPU only: move the waveform to the device "
                                     "(there is no CPU fallback)")
        x = x.float().contiguous()
        B, T = x.shape
        F = self.num_frames(T)
        assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous()
        assert out.shape[0] == B and out.shape[1] >= row_offset + F and out.shape[2] == self.n_mels
        lib = native.load()
        with torch.cuda.device(x.device):
            native.check(lib.m2m_logmel_f32(self._get_plan(), x.data_ptr(), B, T, out.data_ptr(),
                                            out.stride(0), row_offset, native.stream_handle(x.device)),
                         "m2m_logmel_f32")
        return out

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x : waveform(batch, sample) -> (batch, frame, n_mels), as ref input.py:33-41."""
        with torch.no_grad():
            out = torch.empty((x.shape[0], self.num_frames(x.shape[1]), self.n_mels), device=x.device,
                              dtype=torch.float32)
            return self.forward_into(x, out, 0)


class _Holder(nn.Module):
    pass


class _MelSpectrogramBuffers(nn.Module):
    """Holds the two torchaudio buffers under torchaudio's attribute names."""

    def __init__(self, sample_rate: int, n_fft: int, f_min: float, n_mels: int):
        super().__init__()
        self.spectrogram = _Holder()
        self.spectrogram.register_buffer("window", torch.from_numpy(melbank.hann_window(n_fft)))
        self.mel_scale = _Holder()
        self.mel_scale.register_buffer("fb", torch.from_numpy(melbank.mel_filterbank(sample_rate, n_fft, f_min, n_mels)))


class Conditioning(nn.Module):
    """ref: music2midi/input.py:44-59 — embedding rows prepended to the feature sequence."""

    def __init__(self, n_dim: int, num_embeds: list):
        super().__init__()
        self.n_dim = int(n_dim)
        self.embeds = nn.ModuleList([nn.Embedding(num, n_dim) for num in num_embeds])

    def check_indices(self, indices) -> None:
        """Raise nn.Embedding's IndexError for a host-side index outside its table (ref: music2midi/input.py:57)."""
        idx = torch.as_tensor(indices).reshape(-1, len(self.embeds)).long()
        for i, e in enumerate(self.embeds):
            col = idx[:, i]
            if col.numel() and (int(col.min()) < 0 or int(col.max()) >= e.num_embeddings):
                raise IndexError("index out of range in self")

    def write_rows(self, indices: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """Write the len(embeds) conditioning rows of every clip into ``out[:, :n, :]``."""
        n = len(self.embeds)
        if not torch.is_tensor(indices) or not indices.is_cuda:
            # host-side indices (the generate() path hands over a Python list): nn.Embedding's IndexError without a device sync
            # (ref input.py:57).  Device tensors keep the kernel's NaN-row guard instead — checking them would stall the stream.
            indices = torch.as_tensor(indices)
            self.check_indices(indices)
        native.require_gpu()
        indices = indices.to(device=out.device, dtype=torch.long).contiguous()
        assert indices.shape == (out.shape[0], n), f"cond_index must be (batch, {n})"
        tabs = [e.weight.detach() for e in self.embeds]
        for t in tabs:
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in tabs])
        rows = (C.c_int * n)(*[t.shape[0] for t in tabs])
        lib = native.load()
        with torch.cuda.device(out.device):
            native.check(lib.m2m_cond_rows_f32(ptrs, rows, n, self.n_dim, indices.data_ptr(), out.shape[0],
                                               out.data_ptr(), out.stride(0), native.stream_handle(out.device)),
                         "m2m_cond_rows_f32")
        return out

    def forward(self, feature: torch.Tensor, indices: torch.Tensor) -> torch.Tensor:
        """feature (batch, L, n_dim), indices (batch, n_index) -> (batch, n_index + L, n_dim)."""
        n = len(self.embeds)
        B, L, D = feature.shape
        out = torch.empty((B, n + L, D), device=feature.device, dtype=torch.float32)
        out[:, n:, :] = feature
        return self.write_rows(indices, out)
