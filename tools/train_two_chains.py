#!/usr/bin/env python3
"""Experiment: would two half-batches on two streams beat one batch?  Two independent trainers of B/2 clips each (own streams,
own graphs) stepped back to back from one thread, against one trainer of B clips.  python tools/train_two_chains.py [bf16|fp8]"""
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


def make(B, S, Ld):
    model = T5Transformer(cfg.to_dict(), precision="fp32"); load_t5_state(model, sd, strict=False); model = model.cuda()
    tr = NativeTrainer(model, B, S, Ld, precision=prec)
    x = torch.from_numpy(synth.normal(1, "x", (B, S, 384), 2.0)).cuda()
    cond = torch.from_numpy(synth.cond_index_batch(0, B)).cuda()
    labels = (torch.from_numpy((synth.uniform01(4, "l", B * Ld) * 330).astype(np.int64).reshape(B, Ld)) + 3).cuda()
    return tr, x, cond, labels


S, Ld, n = 261, 256, 20
for B in (16, 64):
    one = make(B, S, Ld)
    halves = [make(B // 2, S, Ld) for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    for _ in range(3):
        one[0].forward_backward(*one[1:])
        for h, st in zip(halves, streams):
            with torch.cuda.stream(st):
                h[0].forward_backward(*h[1:])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): one[0].forward_backward(*one[1:])
    torch.cuda.synchronize(); t_one = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n): halves[0][0].forward_backward(*halves[0][1:])
    torch.cuda.synchronize(); t_half = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        for h, st in zip(halves, streams):
            with torch.cuda.stream(st):
                h[0].forward_backward(*h[1:])
    torch.cuda.synchronize(); t_two = (time.perf_counter() - t0) / n
    print(f"{prec} B={B}: one batch {t_one*1e3:.2f} ms; one half alone {t_half*1e3:.2f} ms; two halves on two streams {t_two*1e3:.2f} ms", flush=True)
    one[0].close(); [h[0].close() for h in halves]
