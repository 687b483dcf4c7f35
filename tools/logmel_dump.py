#!/usr/bin/env python3
"""Dump device log-mel outputs for offline error analysis: gpurun_out/logmel_dump.npz"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from music2midi_amd import synth
from music2midi_amd.input import LogMelSpectrogram
out = {}
for kind in ("music", "tones", "noise"):
    for (T, B, nm) in ((4096, 3, 128), (48000, 2, 384), (5000, 5, 384)):
        fe = LogMelSpectrogram(16000, 2048, 256, 20.0, nm)
        wav = torch.from_numpy(synth.waveform_batch(0, B, T, kind)).cuda()
        out[f"{kind}_{T}_{B}_{nm}"] = fe(wav).cpu().numpy()
Path("gpurun_out").mkdir(exist_ok=True)
np.savez_compressed("gpurun_out/logmel_dump.npz", **out)
print("ok", {k: v.shape for k, v in out.items()})
