#!/usr/bin/env python3
"""Phase times of ONE workgroup of resid_panel_kernel (diagnostic build -DM2M_RP_STAMP, loaded through M2M_LIBRARY): shader-clock
stamps at kernel entry, after the prologue and after every chunk's barrier, then after each half of the epilogue."""
import ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from music2midi_amd import native, synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import T5Geometry, default_config
from music2midi_amd.transformer import T5Transformer
cfg = default_config(); geom = T5Geometry(cfg.model.t5); sd = synth.t5_state_dict(geom, 0)
m = T5Transformer(cfg.to_dict(), precision="bf16"); load_t5_state(m, sd, strict=False); m = m.cuda().eval()
x = torch.from_numpy(synth.normal(3, "e", (32, 864, 384), 3.0)).cuda()
for _ in range(3): m._encode(x, 8)
torch.cuda.synchronize()
lib = native.load()
buf = (C.c_ulonglong * 64)()
lib.m2m_debug_rp_stamps.restype = C.c_int
assert lib.m2m_debug_rp_stamps(buf) == 0
v = list(buf)
# the last launch of an encode is the down projection of the last layer (K = 1152: 18 chunks)
t0 = v[0]
print("stamp  cycles-from-entry   delta")
prev = t0
for i, t in enumerate(v):
    if t == 0 or t < t0: continue
    print(f"{i:3d}  {t - t0:10d}  {t - prev:8d}")
    prev = t
