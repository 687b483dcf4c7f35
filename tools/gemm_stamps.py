#!/usr/bin/env python3
"""Phase times of one workgroup of gemm_kernel in the 16-clip training step (diagnostic build -DM2M_GEMM_STAMP; last launch per (EPI, tile)):
   M2M_BUILD_EXTRA=-DM2M_GEMM_STAMP M2M_BUILD_TAG=gstamp python -m music2midi_amd.csrc.build
   M2M_LIBRARY=music2midi_amd/lib/libmusic2midi_amd_gstamp.so M2M_TRAIN_GRAPH=0 python tools/gemm_stamps.py"""
import ctypes as C, sys
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
buf = (C.c_uint64 * 192)()
assert lib.m2m_debug_gemm_stamps(buf) == 0
names = ["entry", "first operand chunks landed", "first k-step done", "k loop done", "epilogue stores landed"]
for epi, what in ((0, "EPI_STORE"), (1, "EPI_RESID"), (4, "EPI_STORE_F32")):
    for tf in (1, 2):
        st = [buf[(epi * 3 + tf) * 8 + i] for i in range(5)]
        if st[0] == 0: continue
        print(f"{what}, {64 * tf}x{64 * tf} tile (last launch of the step with this epilogue)")
        prev = st[0]
        for i in range(1, 5):
            if st[i] == 0: continue
            print(f"   {names[i]:30s} +{(st[i] - prev) * 0.01:6.2f} us   (t = {(st[i] - st[0]) * 0.01:6.2f})")
            prev = st[i]
