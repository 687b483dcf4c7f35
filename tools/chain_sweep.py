#!/usr/bin/env python3
"""Decode throughput over (library build, clips per chain): one process per (build, GPU_MAX_HW_QUEUES) because both
are fixed at process start; M2M_GROUP_ROWS is read per generate call.

    python tools/chain_sweep.py [B]          # uses $M2M_LIBRARY if set
"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from music2midi_amd import synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import T5Geometry, default_config
from music2midi_amd.transformer import T5Transformer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rows_list = [int(r) for r in os.environ.get("SWEEP_ROWS", "32,16,8,4").split(",")]
cfg = default_config(); geom = T5Geometry(cfg.model.t5)
sd = synth.t5_state_dict(geom, 0)
model = T5Transformer(cfg.to_dict(), precision=os.environ.get("SWEEP_PREC", "bf16"))
load_t5_state(model, sd, strict=False)
model = model.cuda().eval()
x = torch.from_numpy(synth.normal(3, "e", (B, 864, 384), 3.0)).cuda()
ref = None
tag = f"lib={Path(os.environ.get('M2M_LIBRARY', 'product')).stem[-12:]} hwq={os.environ.get('GPU_MAX_HW_QUEUES', 'dflt')}"
for rows in rows_list:
    os.environ["M2M_GROUP_ROWS"] = str(rows)
    ids = model.generate_from_embeds(x, max_length=1024)
    if ref is None: ref = ids.clone()
    same = bool(torch.equal(ids, ref))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 3
    for _ in range(n): model.generate_from_embeds(x, max_length=1024)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{tag} B={B} rows/chain={rows:3d}: {dt * 1e3:7.1f} ms  {B * 1023 / dt:9.0f} tok/s  {dt / 1023 * 1e6:6.1f} us/step  ids_same={same}", flush=True)
