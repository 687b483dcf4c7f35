#!/usr/bin/env python3
"""Throughput table over precisions / batch sizes / geometries."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from music2midi_amd import synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import T5Geometry, default_config
from music2midi_amd.input import ModelInputs
from music2midi_amd.transformer import T5Transformer

cfg = default_config(); geom = T5Geometry(cfg.model.t5)
sd = synth.t5_state_dict(geom, 0)
rows = []
for prec in ("bf16", "fp32"):
    model = T5Transformer(cfg.to_dict(), precision=prec)
    load_t5_state(model, sd, strict=False)
    model = model.cuda().eval()
    for (B, T) in ((1, 220500), (4, 220500), (32, 220500), (64, 220500), (128, 48000)):
        if prec == "fp32" and B == 64:
            continue
        wav = torch.from_numpy(synth.waveform_batch(0, B, T)).cuda()
        cond = torch.from_numpy(synth.cond_index_batch(0, B)).cuda()
        inp = ModelInputs(input_waveform=wav, cond_index=cond)
        model.generate(inp, max_length=1024)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        toks = model.generate(inp, max_length=1024)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        n = (toks.shape[1] - 1) * B
        print(f"{prec}  B={B:3d}  T={T:6d} (S={3 + T // 256:3d})  {dt * 1e3:8.1f} ms  {n / dt:10.0f} tok/s  {dt / (toks.shape[1] - 1) * 1e6:7.1f} us/step", flush=True)
