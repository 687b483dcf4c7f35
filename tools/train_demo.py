#!/usr/bin/env python3
"""End to end on synthetic music: train the full model (log-mel frontend -> encoder-decoder, native forward/backward/
Adafactor) on a handful of clips of decaying harmonic tones whose notes are known, then transcribe them back with the
KV-cached greedy decoder and score the result with the chroma metric.  python tools/train_demo.py [steps]"""
import sys, time, copy
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from music2midi_amd import synth
from music2midi_amd.config import DEFAULT_CONFIG
from music2midi_amd.input import ModelInputs
from music2midi_amd.model import Music2MIDI


def make_clip(seed, sr=16000, dur=1.5, n_notes=3):
    u = synth.uniform01(seed, "demo", n_notes * 2).reshape(n_notes, 2)
    t = np.arange(int(sr * dur)) / sr
    y = np.zeros_like(t)
    notes = []
    for i, (pu, du) in enumerate(u):
        on = 0.05 + i * (dur - 0.2) / n_notes
        off = on + 0.25 + 0.15 * du
        pitch = 55 + int(pu * 24)
        f0 = 440.0 * 2 ** ((pitch - 69) / 12)
        env = np.where((t >= on) & (t < off), np.exp(-(t - on) * 4.0), 0.0)
        for h in range(1, 5):
            y += (0.4 / h) * env * np.sin(2 * np.pi * f0 * h * (t - on))
        notes.append([on, off, pitch, 80])
    return (0.8 * y / max(1e-9, np.abs(y).max())).astype(np.float32), np.asarray(notes)


def main(steps=1200, B=8, precision="bf16", verbose=True):
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    cfg["dataloader"]["batch_size"] = B
    cfg["trainer"]["log_every_n_steps"] = 10 ** 9
    m = Music2MIDI(cfg).cuda()
    m.train_precision = precision
    m.eval()                                   # dropout off: the point is to see the machinery learn quickly
    clips = [make_clip(s) for s in range(B)]
    wav = torch.from_numpy(np.stack([c[0] for c in clips])).cuda()
    notes = tuple(c[1] for c in clips)
    idx = torch.from_numpy(synth.cond_index_batch(0, B)).cuda()
    batch = ModelInputs(input_waveform=wav, notes_batch=notes, cond_index=idx)
    from music2midi_amd.evaluation import evaluate_batch
    from music2midi_amd.utils import numpy_to_midi

    def transcribe():     # Music2MIDI.evaluate_batch caps the decode at 4 tokens per label note (ref model.py:57-58), which cuts a
        ids = m.model.generate(batch, max_length=64)          # 3-note clip short (a note costs ~6 tokens); decode freely instead
        out = [numpy_to_midi(n) for n in m.model.tokenizer.decode(ids, mode="batched")]
        return evaluate_batch([numpy_to_midi(n) for n in notes], out), out
    score0 = transcribe()[0]
    (opt,), _ = m.configure_optimizers()
    t0 = time.perf_counter()
    losses = m.fit_batches([batch] * steps, optimizer=opt)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    score1, out_midis = transcribe()
    if verbose:
        print(f"{precision}: {steps} steps in {dt:.1f} s ({1e3 * dt / steps:.2f} ms/step); loss {losses[0]:.3f} -> {losses[-1]:.4f}; "
              f"chroma accuracy of the greedy transcription {score0:.3f} -> {score1:.3f}")
        got = [[int(n.pitch) for n in mm.instruments[0].notes] for mm in out_midis[:4]]
        print("   wanted pitches", [[int(p) for p in n[:, 2]] for n in notes[:4]], "\n   decoded pitches", got)
    return losses, score0, score1


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1200, precision=sys.argv[2] if len(sys.argv) > 2 else "bf16")
