#!/usr/bin/env python3
"""Teacher-forced forward (T5Transformer.forward's device part): batched pass vs KV-cached decode steps.

    python tools/forward_bench.py [B] [Ld]        # M2M_FORWARD=step selects the step-by-step path
"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from music2midi_amd import synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import T5Geometry, default_config
from music2midi_amd.transformer import T5Transformer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
Ld = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg = default_config(); geom = T5Geometry(cfg.model.t5)
for prec in ("bf16", "fp32"):
    model = T5Transformer(cfg.to_dict(), precision=prec)
    load_t5_state(model, synth.t5_state_dict(geom, 0), strict=False)
    model = model.cuda().eval()
    x = torch.from_numpy(synth.normal(1, "e", (B, 864, geom.d_model), 3.0)).cuda()
    ids = torch.from_numpy((synth.uniform01(3, "ids", B * Ld) * 330).astype(np.int64).reshape(B, Ld) + 3).cuda()
    for mode in ("batched", "step"):
        os.environ["M2M_FORWARD"] = mode
        model.logits_from_embeds(x, ids)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = model.logits_from_embeds(x, ids)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{prec} {mode:8s} B={B} S=864 Ld={Ld}: {dt * 1e3:8.1f} ms (encoder included)  logits {tuple(out.shape)}", flush=True)
