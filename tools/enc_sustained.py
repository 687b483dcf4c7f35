#!/usr/bin/env python3
"""Encoder + cross-K/V passes back to back at the headline geometry (B = 32, S = 864, bf16): the per-pass time against the length of
the burst — 10 passes straight after start-up include the first call's set-up, 50 and more give the sustained figure (1.81 ms)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from music2midi_amd import synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import T5Geometry, default_config
from music2midi_amd.transformer import T5Transformer
cfg = default_config(); geom = T5Geometry(cfg.model.t5); sd = synth.t5_state_dict(geom, 0)
m = T5Transformer(cfg.to_dict(), precision="bf16"); load_t5_state(m, sd, strict=False); m = m.cuda().eval()
x = torch.from_numpy(synth.normal(3, "e", (32, 864, 384), 3.0)).cuda()
for n in (10, 50, 200, 500, 500):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): m._encode(x, 8)
    e1.record(); torch.cuda.synchronize()
    print(f"{n:4d} passes back to back: {e0.elapsed_time(e1) / n:.3f} ms per pass")
