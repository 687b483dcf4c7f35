#!/usr/bin/env python3
"""Round 6 (SURVEY K4 below the row-panel gate): encoder + cross-K/V and the teacher-forced pass at SMALL shapes, bf16,
the fused norm+GEMM kernel in its column-group form (M2M_NORM_GEMM_SPLIT=1, default) against rmsnorm_kernel + gemm_kernel
(M2M_NORM_GEMM_SPLIT=0).  The switch is latched per session: a model per leg.   python tools/enc_small_bench.py"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from music2midi_amd import synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import T5Geometry, default_config
from music2midi_amd.transformer import T5Transformer
cfg = default_config(); geom = T5Geometry(cfg.model.t5); sd = synth.t5_state_dict(geom, 0)


def timed(fn, n):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B, S, Ld in ((1, 864, 64), (4, 190, 256), (8, 190, 256), (2, 864, 128), (16, 190, 256), (4, 864, 256), (16, 864, 64)):
    x = torch.from_numpy(synth.normal(3, "e", (B, S, 384), 3.0)).cuda()
    ids = torch.from_numpy((synth.uniform01(3, "ids", B * Ld) * 330).astype(np.int64).reshape(B, Ld) + 3).cuda()
    res = {}
    for rep in range(2):
        for split in ("1", "0"):
            os.environ["M2M_NORM_GEMM_SPLIT"] = split
            m = T5Transformer(cfg.to_dict(), precision="bf16"); load_t5_state(m, sd, strict=False); m = m.cuda().eval()
            res.setdefault(split, []).append((timed(lambda: m._encode(x, 8), 20), timed(lambda: m.logits_from_embeds(x, ids), 10)))
            del m
    f = lambda v: " / ".join(f"{a:.3f}" for a in v)
    print(f"B={B} S={S} ({-(-B * S // 128)} row blocks): encoder + cross-K/V ms split {f([r[0] for r in res['1']])} vs two-kernel {f([r[0] for r in res['0']])};"
          f"  forward(Ld={Ld}) ms split {f([r[1] for r in res['1']])} vs two-kernel {f([r[1] for r in res['0']])}", flush=True)
