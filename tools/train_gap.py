#!/usr/bin/env python3
"""How much of a training step is GPU-busy?  Run under rocprofv3 --kernel-trace; compares summed kernel durations with wall time."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from music2midi_amd import synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import T5Geometry, default_config
from music2midi_amd.training import NativeTrainer
from music2midi_amd.transformer import T5Transformer
cfg = default_config(); geom = T5Geometry(cfg.model.t5)
model = T5Transformer(cfg.to_dict(), precision="fp32"); load_t5_state(model, synth.t5_state_dict(geom, 0), strict=False); model = model.cuda()
B, S, Ld = 16, 261, 256
import os
tr = NativeTrainer(model, B, S, Ld, precision=os.environ.get("M2M_GAP_PREC", "bf16"))
if os.environ.get("M2M_GAP_DROPOUT"): tr.set_dropout(float(os.environ["M2M_GAP_DROPOUT"]), 1)      # e.g. 0.1: the reference's training mode
x = torch.from_numpy(synth.normal(1, "x", (B, S, 384), 2.0)).cuda(); cond = torch.from_numpy(synth.cond_index_batch(0, B)).cuda()
labels = (torch.from_numpy((synth.uniform01(4, "l", B * Ld) * 330).astype(np.int64).reshape(B, Ld)) + 3).cuda()
for _ in range(3): tr.forward_backward(x, cond, labels); tr.optimizer_step()
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20
for _ in range(n): tr.forward_backward(x, cond, labels); tr.optimizer_step()
torch.cuda.synchronize(); print(f"WALL per step {1e3 * (time.perf_counter() - t0) / n:.3f} ms over {n} steps (+3 warm-up)")
