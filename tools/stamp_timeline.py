#!/usr/bin/env python3
"""Diagnostic: per-kernel wall-clock timeline of one decode step from in-kernel stamps.

    M2M_BUILD_VARIANT=stamps python -m music2midi_amd.csrc.build
    M2M_LIBRARY=music2midi_amd/lib/libmusic2midi_amd_stamps.so M2M_GROUP_ROWS=32 python tools/stamp_timeline.py
"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from music2midi_amd import native, synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import T5Geometry, default_config
from music2midi_amd.transformer import T5Transformer

NAMES = {1: "gemm_qkv", 2: "gemm_plain", 3: "gemm_resid", 4: "ff", 6: "attn_cross", 7: "attn_self", 8: "head"}

cfg = default_config()
geom = T5Geometry(cfg.model.t5)
model = T5Transformer(cfg.to_dict(), precision="bf16")
load_t5_state(model, synth.t5_state_dict(geom, 0), strict=False)
model = model.cuda().eval()
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.from_numpy(synth.normal(1, "e", (BATCH, 864, 384), 3.0)).cuda()
lib = native.load()
lib.m2m_debug_read_stamps.restype = C.c_int
lib.m2m_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros(1 << 18, dtype=np.uint64)
model.generate_from_embeds(x, max_length=600)          # warm + reach long self-attention lengths
lib.m2m_debug_read_stamps(buf.ctypes.data, len(buf))   # reset
model.generate_from_embeds(x, max_length=600)
n = lib.m2m_debug_read_stamps(buf.ctypes.data, len(buf))
ev = [(int(v >> 56), int((v >> 48) & 0xFF), int(v & 0xFFFFFFFFFFFF)) for v in buf[:n] if v != 0]
ev.sort(key=lambda e: e[2])
# take one step late in the run: from a head stamp phase 2 to the next head stamp phase 2
heads = [i for i, e in enumerate(ev) if e[0] == 8 and e[1] == 2]
a, b = heads[-3], heads[-2]
seg = ev[a:b + 1]
t0 = seg[0][2]
print(f"{n} stamps; one step = {(seg[-1][2] - t0) * 0.01:.1f} us (100 MHz ticks)")
prev_end = t0
cur = None
rows = []
for kid, ph, tk in seg[1:]:
    if ph == 0:
        cur = [kid, tk, None, None]
        rows.append(cur)
    elif cur is not None and kid == cur[0]:
        if ph == 1:
            cur[2] = tk
        if ph == 2:
            cur.append(tk)
        cur[3] = tk
agg = {}
rows = [r[:4] + [r[4] if len(r) > 4 else None] for r in rows]
for kid, s, mid, e, m2 in rows:
    if e is None:
        continue
    gap = (s - prev_end) * 0.01
    dur = (e - s) * 0.01
    agg.setdefault(kid, []).append((gap, dur, ((mid - s) * 0.01 if mid else 0)))
    prev_end = e
for kid, v in sorted(agg.items()):
    g = np.mean([x[0] for x in v]); d = np.mean([x[1] for x in v]); m = np.mean([x[2] for x in v])
    print(f"{NAMES.get(kid, kid):12s} n={len(v):3d}  gap-before {g:6.2f} us   in-kernel(block0) {d:6.2f} us   to-mid {m:6.2f} us")
print("attention kernels, all phases (us from kernel start): ph4=x arrived, ph5=hn ready, ph6=projection summed, ph1=q ready, ph2=stream consumed, ph7=o ready, ph3=end")
cur = None
shown = 0
for kid, ph, tk in seg[1:]:
    if kid in (4, 6, 7):
        if ph == 0:
            cur = tk
            line = [NAMES[kid]]
        elif cur is not None:
            line.append(f"ph{ph}={(tk - cur) * 0.01:.2f}")
            if ph == (2 if kid == 4 else 3):
                print("  " + "  ".join(line)); shown += 1
                if shown >= 9: break
print("first 20 kernels of the step (gap, dur):")
pe = t0
for kid, s, mid, e, m2 in rows[:14]:
    if e is None:
        continue
    print(f"  {NAMES.get(kid, kid):12s} gap {(s - pe) * 0.01:5.2f}  dur {(e - s) * 0.01:5.2f}  ph1 {((mid - s) * 0.01 if mid else 0):5.2f}  ph2 {((m2 - s) * 0.01 if m2 else 0):5.2f}")
    pe = e
