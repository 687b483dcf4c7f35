"""Deterministic synthetic weights and waveforms.

There is no network for checkpoints or datasets, so the benchmark, the smoke
test and the parity tests all run on random-init weights and synthetic clips
(BASELINE.md §2).  The numbers must be identical in the build container (where
the golden vectors are generated against HuggingFace T5) and on the GPU box, so
they come from a counter-based integer hash written here — not from any
library RNG stream.

Weight scales follow HF T5's ``_init_weights`` with ``initializer_factor=1``
(hf: models/t5/modeling_t5.py:563-616) and 4.34.0 semantics for the untied
``lm_head``; the key layout is the one a Lightning
checkpoint of the reference carries under ``model.*``.
"""
from __future__ import annotations

import zlib
from typing import Dict, Mapping

import numpy as np

from .config import T5Geometry

_U64 = np.uint64
_MASK = (1 << 64) - 1


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """One splitmix64 output per counter value (vectorised, wraps mod 2**64)."""
    with np.errstate(over="ignore"):
        z = x + _U64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        return z ^ (z >> _U64(31))


def _stream_key(seed: int, name: str) -> int:
    h = zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF
    k = ((seed & 0xFFFFFFFF) << 32) | h
    # one scalar splitmix round so nearby seeds give unrelated streams
    z = (k + 0x9E3779B97F4A7C15) & _MASK
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
    return z ^ (z >> 31)


def uniform01(seed: int, name: str, n: int, offset: int = 0) -> np.ndarray:
    """n doubles in [0, 1) from stream (seed, name), starting at counter offset."""
    key = _U64(_stream_key(seed, name))
    with np.errstate(over="ignore"):
        ctr = np.arange(offset, offset + n, dtype=np.uint64) * _U64(0xD1342543DE82EF95) + key
    bits = _splitmix64(ctr)
    return (bits >> _U64(11)).astype(np.float64) * (1.0 / (1 << 53))


def normal(seed: int, name: str, shape, std: float = 1.0) -> np.ndarray:
    """N(0, std^2) float32 tensor via Box-Muller on two uniform streams."""
    n = int(np.prod(shape))
    m = (n + 1) // 2
    u1 = uniform01(seed, name + "#r", m)
    u2 = uniform01(seed, name + "#t", m)
    r = np.sqrt(-2.0 * np.log1p(-u1))  # log1p(-u): u in [0,1) never gives log(0)
    z = np.concatenate([r * np.cos(2.0 * np.pi * u2), r * np.sin(2.0 * np.pi * u2)])[:n]
    return (z * std).astype(np.float32).reshape(shape)


def waveform(clip_index: int, n_samples: int, kind: str = "noise") -> np.ndarray:
    """Synthetic mono clip, float32 in [-1, 1).

    ``noise``  : U[-1, 1) white noise, seed = clip index.
    ``tones``  : three sinusoids with a silent tail (exercises the 1e-6 clamp).
    ``zeros``  : digital silence (every log-mel value must be log(1e-6)).
    ``music``  : piano-like material — ~4 notes per second, each six decaying harmonics of a pitch in
                 MIDI 40..87 with random phases, peak 0.9, and a digitally silent last 15 % (the input
                 class the model actually sees: tonal, wide in-frame dynamic range, silence).
    """
    if kind == "noise":
        return (uniform01(clip_index, "waveform", n_samples) * 2.0 - 1.0).astype(np.float32)
    if kind == "zeros":
        return np.zeros(n_samples, dtype=np.float32)
    if kind == "tones":
        t = np.arange(n_samples, dtype=np.float64)
        sr = 16000.0
        f = [220.0 * (1 + clip_index % 3), 1318.5, 3520.0]
        y = 0.5 * np.sin(2 * np.pi * f[0] * t / sr) + 0.3 * np.sin(2 * np.pi * f[1] * t / sr)
        y += 0.15 * np.sin(2 * np.pi * f[2] * t / sr + 0.7)
        y[(n_samples * 3) // 4:] = 0.0
        return y.astype(np.float32)
    if kind == "music":
        sr = 16000.0
        t = np.arange(n_samples, dtype=np.float64) / sr
        n_notes = max(3, int(n_samples / sr * 4))
        u = uniform01(clip_index, "music", n_notes * 9).reshape(n_notes, 9)
        y = np.zeros(n_samples, dtype=np.float64)
        for on_u, pitch_u, dur_u, *phases in u:
            on = on_u * (n_samples / sr) * 0.8
            f0 = 440.0 * 2.0 ** ((40 + int(pitch_u * 48) - 69) / 12.0)
            dur = 0.2 + dur_u
            env = np.where((t >= on) & (t < on + dur), np.exp(-(t - on) / (dur / 3.0)), 0.0)
            for h in range(1, 7):
                if f0 * h < sr / 2:
                    y += (0.3 / h) * env * np.sin(2.0 * np.pi * f0 * h * (t - on) + 2.0 * np.pi * phases[h - 1])
        y[int(n_samples * 0.85):] = 0.0
        peak = np.abs(y).max()
        if peak > 0.9:
            y *= 0.9 / peak
        return y.astype(np.float32)
    raise ValueError(f"unknown waveform kind {kind!r}")


def waveform_batch(first_clip: int, batch: int, n_samples: int, kind: str = "noise") -> np.ndarray:
    return np.stack([waveform(first_clip + i, n_samples, kind) for i in range(batch)])


def cond_index_batch(first_clip: int, batch: int, sizes=(6, 3)) -> np.ndarray:
    """cond_index = (clip mod 6, clip mod 3)."""
    idx = np.arange(first_clip, first_clip + batch, dtype=np.int64)
    return np.stack([idx % s for s in sizes], axis=1)


def t5_state_dict(geom: T5Geometry, seed: int = 0, cond_sizes=(6, 3)) -> Dict[str, np.ndarray]:
    """Random-init weights in the reference checkpoint key layout (minus 'model.').

    Keys: ``transformer.*`` (HF T5 names, separate ``lm_head.weight``) and
    ``conditioning.embeds.{i}.weight``.
    """
    d, dff, inner, H = geom.d_model, geom.d_ff, geom.inner_dim, geom.num_heads
    dk, V = geom.d_kv, geom.vocab_size
    sd: Dict[str, np.ndarray] = {}

    def put(name, shape, std):
        sd[name] = normal(seed, name, shape, std)

    def ones(name, n):
        sd[name] = np.ones(n, dtype=np.float32)

    def attn(prefix, has_bias):
        put(f"{prefix}.q.weight", (inner, d), (d * dk) ** -0.5)
        put(f"{prefix}.k.weight", (inner, d), d ** -0.5)
        put(f"{prefix}.v.weight", (inner, d), d ** -0.5)
        put(f"{prefix}.o.weight", (d, inner), inner ** -0.5)
        if has_bias:
            put(f"{prefix}.relative_attention_bias.weight", (geom.num_buckets, H), d ** -0.5)

    def ffn(prefix):
        put(f"{prefix}.wi_0.weight", (dff, d), d ** -0.5)
        put(f"{prefix}.wi_1.weight", (dff, d), d ** -0.5)
        put(f"{prefix}.wo.weight", (d, dff), dff ** -0.5)

    put("transformer.shared.weight", (V, d), 1.0)
    for i in range(geom.num_layers):
        p = f"transformer.encoder.block.{i}"
        attn(f"{p}.layer.0.SelfAttention", i == 0)
        ones(f"{p}.layer.0.layer_norm.weight", d)
        ffn(f"{p}.layer.1.DenseReluDense")
        ones(f"{p}.layer.1.layer_norm.weight", d)
    ones("transformer.encoder.final_layer_norm.weight", d)
    for i in range(geom.num_decoder_layers):
        p = f"transformer.decoder.block.{i}"
        attn(f"{p}.layer.0.SelfAttention", i == 0)
        ones(f"{p}.layer.0.layer_norm.weight", d)
        attn(f"{p}.layer.1.EncDecAttention", False)
        ones(f"{p}.layer.1.layer_norm.weight", d)
        ffn(f"{p}.layer.2.DenseReluDense")
        ones(f"{p}.layer.2.layer_norm.weight", d)
    ones("transformer.decoder.final_layer_norm.weight", d)
    put("transformer.lm_head.weight", (V, d), 1.0)
    for i, n in enumerate(cond_sizes):
        put(f"conditioning.embeds.{i}.weight", (n, d), 1.0)
    return sd


def perturb_layer_norms(sd: Dict[str, np.ndarray], seed: int = 0, amount: float = 0.25) -> None:
    """Make every RMSNorm weight non-trivial (1 + amount*N(0,1)) in place.

    Random init leaves them at exactly 1, which would hide a kernel that forgot
    to apply them; the parity fixtures therefore use perturbed norms.
    """
    for k in list(sd):
        if k.endswith("layer_norm.weight"):
            sd[k] = (1.0 + amount * normal(seed, k + "#ln", sd[k].shape, 1.0)).astype(np.float32)


def force_eos_head(sd: Mapping[str, np.ndarray], geom: T5Geometry, active: int = 12,
                   eos_scale: float = 0.55) -> None:
    """Craft an ``lm_head`` under which greedy rows hit EOS at different steps.

    Random weights never emit EOS, leaving pad-after-EOS and
    the early stop unpinned.  Zeroing all but ``active`` rows of the head makes
    the argmax a race among few ids; scaling the EOS row sets how long that
    race usually lasts.
    """
    w = sd["transformer.lm_head.weight"]
    w[active:, :] = 0.0
    w[geom.eos_token_id, :] *= eos_scale
    w[geom.pad_token_id, :] = 0.0
