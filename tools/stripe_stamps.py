#!/usr/bin/env python3
"""Phase times of one workgroup of attn_stripe_kernel in the 16-clip training step (diagnostic build -DM2M_ST_STAMP):
   M2M_BUILD_EXTRA=-DM2M_ST_STAMP M2M_BUILD_TAG=ststamp python -m music2midi_amd.csrc.build
   M2M_LIBRARY=music2midi_amd/lib/libmusic2midi_amd_ststamp.so M2M_TRAIN_GRAPH=0 python tools/stripe_stamps.py"""
import ctypes as C, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from music2midi_amd import native, synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import T5Geometry, default_config
from music2midi_amd.training import NativeTrainer
from music2midi_amd.transformer import T5Transformer
cfg = default_config(); geom = T5Geometry(cfg.model.t5)
model = T5Transformer(cfg.to_dict(), precision="fp32"); load_t5_state(model, synth.t5_state_dict(geom, 0), strict=False); model = model.cuda()
B, S, Ld = 16, 261, 256
tr = NativeTrainer(model, B, S, Ld, precision="bf16")
x = torch.from_numpy(synth.normal(1, "x", (B, S, 384), 2.0)).cuda(); cond = torch.from_numpy(synth.cond_index_batch(0, B)).cuda()
labels = (torch.from_numpy((synth.uniform01(4, "l", B * Ld) * 330).astype(np.int64).reshape(B, Ld)) + 3).cuda()
for _ in range(3): tr.forward_backward(x, cond, labels)
torch.cuda.synchronize()
lib = native.load()
buf = (C.c_uint64 * 48)()
assert lib.m2m_debug_stripe_stamps(buf) == 0
names = ["entry", "operand fragments landed", "P rows / bias in LDS", "pass 1 done", "row statistics merged", "pass 2 done", "rows staged (barrier)",
         "rows copied out", "diagonal sums", "fused product + end"]
for v, what in enumerate(["forward, no bias (last launch: decoder cross, layer 5)", "forward, bias (encoder self / decoder self)", "backward (last launch: encoder self, layer 0)", "-"]):
    st = [buf[v * 12 + i] for i in range(12)]
    if st[0] == 0: continue
    print(what)
    prev = st[0]
    for i in range(1, 10):
        if st[i] == 0: continue
        print(f"   {names[i]:28s} +{(st[i] - prev) * 0.01:6.2f} us   (t = {(st[i] - st[0]) * 0.01:6.2f})")
        prev = st[i]
