"""GPU: the callers either side of the path, through the reference's own entry points —
Music2MIDI.evaluate_batch / validation_step (ref model.py:45-65) and
Music2MIDI.load_from_checkpoint(<Lightning-shaped .ckpt>).cuda().generate(audio_y=...) (ref evaluate.py:27,43)."""
import copy

import numpy as np
import pytest
import torch

from music2midi_amd import synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import DEFAULT_CONFIG, T5Geometry
from music2midi_amd.input import ModelInputs

from lightning_ckpt import write_lightning_like_ckpt

pytestmark = pytest.mark.gpu


def _model_and_oracle(eos=True):
    from music2midi_amd.model import Music2MIDI
    from oracle.t5 import T5Oracle
    geom = T5Geometry(DEFAULT_CONFIG["model"]["t5"])
    sd = synth.t5_state_dict(geom, seed=0)
    synth.perturb_layer_norms(sd, 0)
    if eos:
        synth.force_eos_head(sd, geom, active=340, eos_scale=1.6)
    m = Music2MIDI(copy.deepcopy(DEFAULT_CONFIG))
    load_t5_state(m.model, sd, strict=False)
    return m, sd, geom, T5Oracle(geom, sd)


def _oracle_inputs(sd, wav, idx):
    from oracle.logmel import LogMelOracle, conditioning
    emb = [torch.from_numpy(sd[f"conditioning.embeds.{i}.weight"]) for i in range(2)]
    return conditioning(LogMelOracle(16000, 2048, 256, 20.0, 384)(wav), idx, emb)


def _label_notes():
    a = np.array([[0.10, 0.40, 60, 80], [0.50, 1.00, 64, 80], [1.20, 1.90, 67, 80], [2.00, 2.60, 72, 80]])
    b = np.array([[0.05, 0.30, 50, 80], [0.70, 1.10, 55, 80]])
    c = np.array([[0.00, 2.90, 40, 80], [0.30, 0.80, 76, 80], [1.00, 1.40, 77, 80], [1.50, 1.70, 79, 80],
                  [2.00, 2.20, 81, 80], [2.40, 2.80, 83, 80]])
    return (a, b, c)


def test_evaluate_batch_and_validation_step_match_the_oracle_pipeline():
    from music2midi_amd.utils import numpy_to_midi
    from oracle import chroma
    m, sd, geom, orc = _model_and_oracle()
    m = m.cuda().eval()
    notes = _label_notes()
    B, T = len(notes), 48000
    wav = torch.from_numpy(synth.waveform_batch(40, B, T))
    idx = torch.from_numpy(synth.cond_index_batch(40, B))
    inputs = ModelInputs(input_waveform=wav.cuda(), notes_batch=notes, cond_index=idx.cuda())
    score, out_midis, label_midis = m.evaluate_batch(inputs)
    # the oracle's version of the same pipeline: max_length = 4 * (most notes in a clip) = 24 (ref model.py:57-58)
    x_ref = _oracle_inputs(sd, wav, idx)
    ids_ref = orc.generate(x_ref, 4 * max(len(n) for n in notes))
    assert ids_ref.shape[1] <= 24
    ids_dev = m.model.generate(inputs, max_length=24).cpu()
    assert torch.equal(ids_dev, ids_ref)
    dec_ref = m.model.tokenizer.decode(ids_ref, mode="batched")
    assert len(out_midis) == len(label_midis) == B
    for midi, want in zip(out_midis, dec_ref):
        got = np.array([[n.start, n.end, n.pitch, n.velocity] for n in midi.instruments[0].notes]).reshape(-1, 4)
        want = np.asarray(want, dtype=np.float64).reshape(-1, 4)
        want = want[want[:, 1] > want[:, 0]]                      # numpy_to_midi drops invalid notes (ref utils.py:19)
        assert np.array_equal(got, want)
    assert score == pytest.approx(chroma.evaluate_batch(list(notes), [np.asarray(d).reshape(-1, 4) for d in dec_ref]), abs=1e-12)
    # validation_step: returns the teacher-forced LOSS (a tensor, as the reference), logs loss + score
    loss = m.validation_step(inputs, 0)
    labels = m.model.tokenizer(notes)
    labels[labels == 0] = -100
    loss_ref, _ = orc.forward(x_ref, labels)
    assert torch.is_tensor(loss) and loss.dim() == 0 and abs(loss.item() - loss_ref.item()) < 1e-4
    assert m.logged["val/score"] == pytest.approx(score) and m.logged["val/loss"] == pytest.approx(loss.item())
    assert m.logged["batch_size"] == B


def test_lightning_checkpoint_to_generate_matches_oracle(tmp_path):
    """ref evaluate.py:27,43 — load_from_checkpoint(ckpt, config_path=...).cuda(); model.eval(); model.generate(...)."""
    from music2midi_amd.model import Music2MIDI
    src, sd, geom, orc = _model_and_oracle()
    ck = write_lightning_like_ckpt(tmp_path / "epoch=412-step=51912.ckpt", src.state_dict())
    del src
    model = Music2MIDI.load_from_checkpoint(str(ck), config_path=copy.deepcopy(DEFAULT_CONFIG)).cuda()
    model.eval()
    seg = 48000
    audio = synth.waveform(33, 2 * seg - 700)
    notes = model.generate_notes(audio_y=audio, cond_index=[4, 2])
    padded = np.pad(audio, (0, 2 * seg - len(audio)))
    rows = []
    for i in range(2):
        x = _oracle_inputs(sd, torch.from_numpy(padded[i * seg:(i + 1) * seg])[None], torch.tensor([[4, 2]]))
        rows.append(orc.generate(x, 1024)[0])
    want = model.model.tokenizer.decode(rows, mode="sequential", duration_per_batch=3)
    assert notes.shape == want.shape and np.array_equal(notes, want)
    midi = model.generate(audio_y=audio, cond_index=[4, 2])
    assert len(midi.instruments[0].notes) == int((want[:, 1] > want[:, 0]).sum())


def test_generate_from_a_float_wav_file(tmp_path):
    """audio_path ingest (ref model.py:83-84) without librosa: 32-bit float stereo WAV at 44.1 kHz -> model rate."""
    import struct
    from music2midi_amd.audio import load_audio
    m, sd, geom, orc = _model_and_oracle()
    m = m.cuda().eval()
    sr_in = 44100
    y = synth.waveform(5, sr_in * 2) * 0.5
    stereo = np.stack([y, y], axis=1).astype("<f4")
    fmt = struct.pack("<HHIIHH", 3, 2, sr_in, sr_in * 8, 8, 32)
    body = b"fmt " + struct.pack("<I", 16) + fmt + b"data" + struct.pack("<I", stereo.nbytes) + stereo.tobytes()
    p = tmp_path / "clip.wav"
    p.write_bytes(b"RIFF" + struct.pack("<I", 4 + len(body)) + b"WAVE" + body)
    notes = m.generate_notes(audio_path=p, cond_index=[1, 0])
    audio = load_audio(p, 16000)
    assert abs(len(audio) - 32000) <= 1
    want = m.generate_notes(audio_y=audio, cond_index=[1, 0])
    assert np.array_equal(notes, want)
