#!/usr/bin/env python3
"""Encoder passes at several batch sizes (run under rocprofv3 --kernel-trace, then tools/trace_by_grid.py ... attn_): how the
attention kernel's time grows with workgroups per CU (B = 4: 224 workgroups, at most one per CU; B = 32: 7 per CU)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from music2midi_amd import synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import T5Geometry, default_config
from music2midi_amd.transformer import T5Transformer
cfg = default_config(); geom = T5Geometry(cfg.model.t5); sd = synth.t5_state_dict(geom, 0)
m = T5Transformer(cfg.to_dict(), precision="bf16"); load_t5_state(m, sd, strict=False); m = m.cuda().eval()
import os
for B in [int(b) for b in os.environ.get("OCC_BATCHES", "4,8,12,16,24,32").split(",")]:
    x = torch.from_numpy(synth.normal(3, "e", (B, 864, 384), 3.0)).cuda()
    for _ in range(4): m._encode(x, 8)
    torch.cuda.synchronize()
