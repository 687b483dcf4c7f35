#!/usr/bin/env python3
"""Decode time of SMALL batches against the number of chains they are split into (M2M_GROUP_ROWS): does a batch below 24 clips —
or the live rows left after re-packing — decode faster as two or four overlapping chains than as one?  One process per setting
(the switch is read when a session is planned).   python tools/small_batch_chains.py"""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
CODE = r'''
import sys, time, torch
sys.path.insert(0, %r)
from music2midi_amd import synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import DEFAULT_CONFIG, T5Geometry, load_config
from music2midi_amd.transformer import T5Transformer
g = T5Geometry(load_config(DEFAULT_CONFIG).model.t5)
sd = synth.t5_state_dict(g, seed=0)
m = T5Transformer(DEFAULT_CONFIG, precision="bf16"); load_t5_state(m, sd, strict=False); m = m.cuda().eval()
B = int(sys.argv[1])
x = torch.from_numpy(synth.normal(5, "e", (B, 864, g.d_model), 3.0)).cuda()
m.generate_from_embeds(x, max_length=1024)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): out = m.generate_from_embeds(x, max_length=1024)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"{dt * 1e3:8.1f} ms per batch, {(out.shape[1] - 1)} steps -> {dt / (out.shape[1] - 1) * 1e6:6.1f} us per step")
''' % str(ROOT)
for B in (4, 8, 16, 32):
    for rows in sorted({B, max(1, B // 2), max(1, B // 4)}, reverse=True):
        env = dict(os.environ, M2M_GROUP_ROWS=str(rows), M2M_COMPACT="0")
        r = subprocess.run([sys.executable, "-c", CODE, str(B)], env=env, capture_output=True, text=True, timeout=300)
        print(f"B = {B:2d}, chains of {rows:2d} rows ({-(-B // rows)} chains): {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
