#!/usr/bin/env python3
"""Host time of one m2m_train_forward_backward call (enqueue only) against the step's GPU time.  python tools/train_host_time.py [bf16|fp8]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from music2midi_amd import synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import T5Geometry, default_config
from music2midi_amd.training import NativeTrainer
from music2midi_amd.transformer import T5Transformer

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
cfg = default_config(); geom = T5Geometry(cfg.model.t5)
sd = synth.t5_state_dict(geom, 0)
model = T5Transformer(cfg.to_dict(), precision="fp32"); load_t5_state(model, sd, strict=False); model = model.cuda()
S, Ld, n = 261, 256, 20
for B in (4, 8, 16, 64):
    tr = NativeTrainer(model, B, S, Ld, precision=prec)
    x = torch.from_numpy(synth.normal(1, "x", (B, S, 384), 2.0)).cuda()
    cond = torch.from_numpy(synth.cond_index_batch(0, B)).cuda()
    labels = (torch.from_numpy((synth.uniform01(4, "l", B * Ld) * 330).astype(np.int64).reshape(B, Ld)) + 3).cuda()
    for _ in range(3): tr.forward_backward(x, cond, labels)
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for _ in range(n):
        h0 = time.perf_counter(); tr.forward_backward(x, cond, labels); host.append(time.perf_counter() - h0)
    torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / n
    lat = []
    for _ in range(5):
        torch.cuda.synchronize(); h0 = time.perf_counter(); tr.forward_backward(x, cond, labels); torch.cuda.synchronize(); lat.append(time.perf_counter() - h0)
    host.sort(); lat.sort()
    print(f"{prec} B={B}: back-to-back {t_all*1e3:.2f} ms/step; host enqueue median {host[n//2]*1e3:.2f} ms (min {host[0]*1e3:.2f}); "
          f"single synced step {lat[2]*1e3:.2f} ms", flush=True)
    tr.close()
