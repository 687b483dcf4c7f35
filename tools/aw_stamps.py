#!/usr/bin/env python3
"""Phase times of wave 0 of ONE workgroup of attn_wide_kernel (diagnostic build -DM2M_AW_STAMP, loaded through M2M_LIBRARY):
shader-clock stamps (100 MHz s_memtime ticks are NOT shader cycles: the table prints ticks and the share of the whole wave)."""
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
lib.m2m_debug_aw_stamps.restype = C.c_int
assert lib.m2m_debug_aw_stamps(buf) == 0
v = list(buf)
names = {0: "entry", 1: "prologue done (bias table, Q, tile 0 in LDS, tile 1 in flight)", 40: "wide loop done", 41: "masked tail tile done", 42: "stored"}
ph = ["step top", "staged next tile + prefetch issued", "bias + K reads + 8 QK MFMAs issued", "max + exchange + alpha", "exponentials + sums", "rescale + pack + V reads + 8 PV MFMAs issued", "barrier passed"]
for k in range(4):
    for j, n in enumerate(ph): names[2 + 8 * k + j] = f"step {4 + k}: {n}"
t0 = v[0]; prev = t0; total = v[42] - t0
print(f"whole wave: {total} ticks")
for i in sorted(names):
    t = v[i]
    if t == 0: continue
    print(f"{i:3d}  {t - t0:8d}  +{t - prev:6d}  {100.0 * (t - prev) / total:5.1f} %  {names[i]}")
    prev = t
