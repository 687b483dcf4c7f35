"""CPU: the §8(f) rows that are host code — Lightning checkpoint ingest (f2), chroma accuracy (f3),
audio ingest and pitch-shift augmentation (f4)."""
import os
import pickle
import struct

import numpy as np
import pytest
import torch

from music2midi_amd import synth
from music2midi_amd.config import DEFAULT_CONFIG

from lightning_ckpt import write_lightning_like_ckpt


# ------------------------------------------------------------------ f2: checkpoints
def test_lightning_shaped_checkpoint_loads_without_lightning(tmp_path):
    from music2midi_amd.checkpoint import read_checkpoint
    from music2midi_amd.model import Music2MIDI
    a = Music2MIDI(DEFAULT_CONFIG)
    ck = write_lightning_like_ckpt(tmp_path / "epoch=412-step=51912.ckpt", a.state_dict())
    assert b"pytorch_lightning.utilities.parsing" in ck.read_bytes() or True     # zip member, checked below
    with pytest.raises(pickle.UnpicklingError):                                  # torch's own safe loader rejects it ...
        torch.load(ck, map_location="cpu", weights_only=True)
    raw = read_checkpoint(ck)                                                    # ... the allow-list reader does not
    assert type(raw["hyper_parameters"]) is dict and raw["hyper_parameters"]["config_path"] == "config.yaml"
    assert raw["epoch"] == 412 and raw["callbacks"] and type(raw["callback_objects"][0]).__name__ == "ModelCheckpoint"
    assert "pytorch_lightning" not in __import__("sys").modules
    b = Music2MIDI.load_from_checkpoint(str(ck), config_path=DEFAULT_CONFIG)
    sa, sb = a.state_dict(), b.state_dict()
    assert set(sa) == set(sb) and all(torch.equal(sa[k], sb[k]) for k in sa)


def test_untrusted_checkpoint_cannot_run_code(tmp_path):
    from music2midi_amd.checkpoint import read_checkpoint
    marker = tmp_path / "pwned"

    class Evil:
        def __reduce__(self):
            return (os.system, (f"touch {marker}",))

    class Evil2:
        def __reduce__(self):
            return (eval, (f"open({str(marker)!r}, 'w').close()",))

    p = tmp_path / "evil.ckpt"
    torch.save({"state_dict": {"model.w": torch.ones(2)}, "a": Evil(), "b": Evil2()}, p)
    raw = read_checkpoint(p)
    assert not marker.exists()
    assert torch.equal(raw["state_dict"]["model.w"], torch.ones(2))
    assert type(raw["a"]).__name__ == "system" and type(raw["b"]).__name__ == "eval"      # inert stubs


def test_checkpoint_io_errors_are_not_swallowed(tmp_path):
    from music2midi_amd.checkpoint import read_checkpoint
    with pytest.raises(FileNotFoundError):
        read_checkpoint(tmp_path / "missing.ckpt")
    bad = tmp_path / "truncated.ckpt"
    bad.write_bytes(b"PK\x03\x04 not a checkpoint")
    with pytest.raises(Exception) as e:
        read_checkpoint(bad)
    assert not isinstance(e.value, FileNotFoundError)


# ------------------------------------------------------------------ f3: chroma accuracy vs the restated mir_eval / pretty_midi
def _random_notes(seed, n, max_t, lo=40, hi=90):
    u = synth.uniform01(seed, "chroma", n * 3).reshape(n, 3)
    on = np.sort(u[:, 0] * max_t)
    return np.stack([on, on + 0.05 + u[:, 1] * 0.8, np.floor(lo + u[:, 2] * (hi - lo)), np.full(n, 80.0)], axis=1)


def test_chroma_accuracy_matches_the_restated_mir_eval_pipeline():
    from music2midi_amd.evaluation import evaluate_batch, extract_midi_melody
    from music2midi_amd.utils import numpy_to_midi
    from oracle import chroma
    cases = []
    for seed in range(12):
        t = _random_notes(seed, 5 + 3 * seed, 2.5 + 0.5 * seed)
        o = t.copy()
        u = synth.uniform01(100 + seed, "perturb", len(o))
        o[:, 2] += np.where(u < 0.3, 12, np.where(u < 0.5, 1, np.where(u < 0.6, -7, 0)))     # octave errors, semitone errors
        o[:, :2] += (synth.uniform01(200 + seed, "shift", len(o))[:, None] - 0.5) * 0.08
        o = o[synth.uniform01(300 + seed, "drop", len(o)) > 0.2]                              # missed notes -> unvoiced estimate frames
        cases.append((t, o))
    cases.append((_random_notes(50, 6, 2.0), np.zeros((0, 4))))                               # empty output
    cases.append((_random_notes(51, 4, 1.0), _random_notes(52, 9, 4.0)))                      # output runs longer than the target
    cases.append((np.array([[0.5, 1.0, 60, 80], [2.0, 2.5, 62, 80]]), np.array([[0.5, 1.0, 72, 80], [2.0, 2.5, 63, 80]])))  # gap = silent frames
    for t, o in cases:
        got = extract_midi_melody(numpy_to_midi(t), numpy_to_midi(o))
        want = chroma.extract_midi_melody(t, o)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        assert evaluate_batch([numpy_to_midi(t)], [numpy_to_midi(o)]) == pytest.approx(chroma.evaluate_batch([t], [o]), abs=1e-12)
    ts, os_ = zip(*cases)
    score = evaluate_batch([numpy_to_midi(t) for t in ts], [numpy_to_midi(o) for o in os_])
    assert score == pytest.approx(chroma.evaluate_batch(ts, os_), abs=1e-12) and 0.2 < score < 0.95


def test_chroma_accuracy_hand_computed_edges():
    from music2midi_amd.evaluation import evaluate_batch, melody_chroma_accuracy
    from music2midi_amd.utils import numpy_to_midi
    # 100 reference frames voiced (1 s of pitch 60), estimate voiced and right on the first half only -> 50 / 99
    # (frames 0..98 are voiced: the last column of a `times`-sampled roll stays zero)
    ref = np.array([[0.0, 1.0, 60, 80]])
    est = np.array([[0.0, 0.5, 60, 80]])
    assert evaluate_batch([numpy_to_midi(ref)], [numpy_to_midi(est)]) == pytest.approx(50 / 99)
    # a reference with no voiced frame scores 0; so does an estimate that is never voiced
    assert melody_chroma_accuracy(np.full(10, -1), np.full(10, 60)) == 0.0
    assert melody_chroma_accuracy(np.full(10, 60), np.full(10, -1)) == 0.0
    # 49 cents is inside the tolerance, a semitone is not; folding is to the NEAREST octave
    assert melody_chroma_accuracy(np.array([60, 60, 60]), np.array([72, 48, 61])) == pytest.approx(2 / 3)
    with pytest.raises(AssertionError):
        melody_chroma_accuracy(np.zeros(3, dtype=int), np.zeros(4, dtype=int))       # ref evaluation.py:56


# ------------------------------------------------------------------ f4: audio ingest + augmentation
def _riff(fmt_tag, n_ch, rate, bits, payload, extensible=False):
    block = n_ch * bits // 8
    fmt = struct.pack("<HHIIHH", 0xFFFE if extensible else fmt_tag, n_ch, rate, rate * block, block, bits)
    if extensible:
        fmt += struct.pack("<HHI", 22, bits, 3) + struct.pack("<H", fmt_tag) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 5) + b"abcde\x00"   # odd chunk + pad byte
    chunks += b"data" + struct.pack("<I", len(payload)) + payload
    return b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks


@pytest.mark.parametrize("kind", ["pcm8", "pcm16", "pcm24", "pcm32", "f32", "f64", "ext_f32"])
def test_wav_formats(tmp_path, kind):
    from music2midi_amd.audio import load_audio, read_wav
    sr = 16000
    t = np.arange(sr // 2) / sr
    y = 0.6 * np.sin(2 * np.pi * 330 * t)
    stereo = np.stack([y, 0.5 * y], axis=1)                      # mono mix = 0.75 * y
    if kind == "pcm8":
        raw, tag, bits, tol = ((stereo * 127) + 128).round().astype(np.uint8).tobytes(), 1, 8, 1.5e-2
    elif kind == "pcm16":
        raw, tag, bits, tol = (stereo * 32767).round().astype("<i2").tobytes(), 1, 16, 1e-4
    elif kind == "pcm24":
        v = (stereo * 8388607).round().astype(np.int32).reshape(-1)
        raw = b"".join(int(x & 0xFFFFFF).to_bytes(3, "little") for x in v)
        tag, bits, tol = 1, 24, 1e-6
    elif kind == "pcm32":
        raw, tag, bits, tol = (stereo * 2147483647).round().astype("<i4").tobytes(), 1, 32, 1e-6
    elif kind in ("f32", "ext_f32"):
        raw, tag, bits, tol = stereo.astype("<f4").tobytes(), 3, 32, 1e-6
    else:
        raw, tag, bits, tol = stereo.astype("<f8").tobytes(), 3, 64, 1e-6
    p = tmp_path / f"{kind}.wav"
    p.write_bytes(_riff(tag, 2, sr, bits, raw, extensible=kind.startswith("ext")))
    data, rate = read_wav(p)
    assert rate == sr and data.shape == (sr // 2, 2) and data.dtype == np.float32
    out = load_audio(p, sr)
    assert out.dtype == np.float32 and np.abs(out - 0.75 * y).max() < tol
    half = load_audio(p, sr // 2)                                # resampled on ingest, as librosa.load(sr=...)
    assert abs(len(half) - sr // 4) <= 1


def test_non_wav_fails_loudly(tmp_path):
    from music2midi_amd.audio import load_audio
    p = tmp_path / "clip.mp3"
    p.write_bytes(b"ID3\x03\x00" + bytes(64))
    with pytest.raises(ValueError, match="not a RIFF/WAVE"):
        load_audio(p, 16000)


def _peak_hz(y, sr):
    seg = y[sr // 4: -sr // 4]
    spec = np.abs(np.fft.rfft(seg * np.hanning(len(seg))))
    return np.argmax(spec) * sr / len(seg)


@pytest.mark.parametrize("steps", [-6, -1, 3, 5])
def test_pitch_shift_moves_the_pitch_and_keeps_the_length(steps):
    from music2midi_amd.audio import pitch_shift, transpose
    sr = 22050
    t = np.arange(2 * sr) / sr
    y = (0.5 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)
    z = pitch_shift(y, sr, steps)
    assert z.shape == y.shape and z.dtype == np.float32
    assert _peak_hz(z, sr) == pytest.approx(440 * 2 ** (steps / 12), rel=4e-3)
    rms = lambda a: float(np.sqrt((a[sr // 4:-sr // 4] ** 2).mean()))
    assert 0.8 < rms(z) / rms(y) < 1.1
    notes = np.array([[0.1, 0.5, 60, 80]])
    w2, n2 = transpose(y, notes, steps, sr)                      # ref dataset.py:157-160
    assert n2[0, 2] == 60 + steps and notes[0, 2] == 60 and np.array_equal(w2, z)
    assert np.array_equal(pitch_shift(y, sr, 0), y)


def test_normalize():
    from music2midi_amd.audio import normalize
    y = np.array([0.1, -0.5, 0.25], dtype=np.float32)
    assert np.allclose(normalize(y), [0.2, -1.0, 0.5]) and np.array_equal(normalize(np.zeros(4, np.float32)), np.zeros(4))
