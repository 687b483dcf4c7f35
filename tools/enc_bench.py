#!/usr/bin/env python3
"""Encoder + cross-K/V time at the headline geometry (B=32, S=864), both precisions."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from music2midi_amd import synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import T5Geometry, default_config
from music2midi_amd.transformer import T5Transformer
cfg = default_config(); geom = T5Geometry(cfg.model.t5); sd = synth.t5_state_dict(geom, 0)
for prec in ("bf16", "fp32"):
    m = T5Transformer(cfg.to_dict(), precision=prec); load_t5_state(m, sd, strict=False); m = m.cuda().eval()
    x = torch.from_numpy(synth.normal(3, "e", (32, 864, 384), 3.0)).cuda()
    for _ in range(3): m._encode(x, 8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): m._encode(x, 8)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{prec}: encoder + cross-K/V {ms:.3f} ms = {32 * 35.17e9 / ms / 1e9:.0f} TFLOP/s")
