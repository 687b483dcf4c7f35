#!/usr/bin/env python3
"""Soak test: the same batch decoded over and over (several shapes, both precision modes, with and without EOS)
must give bit-identical token ids every time - the integer-atomic residual stream and the three-buffer rotation
make the result independent of workgroup arrival order.  ~1.5 min on an MI355X:  python tools/soak.py"""
import sys, time, copy
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from music2midi_amd import synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import DEFAULT_CONFIG, T5Geometry, load_config
from music2midi_amd.transformer import T5Transformer

def build(cfg, prec, eos):
    g = T5Geometry(load_config(cfg).model.t5)
    sd = synth.t5_state_dict(g, seed=0); synth.perturb_layer_norms(sd, 0)
    if eos: synth.force_eos_head(sd, g)
    m = T5Transformer(cfg, precision=prec); load_t5_state(m, sd, strict=False)
    return m.cuda().eval(), g

bad = 0
for prec in ("bf16", "fp32"):
    for eos in (False, True):
        m, g = build(DEFAULT_CONFIG, prec, eos)
        for (B, S, L) in ((32, 864, 200), (140, 40, 64), (33, 190, 120), (7, 300, 90)):
            x = torch.from_numpy(synth.normal(B, "e", (B, S, g.d_model), 3.0)).cuda()
            ref = m.generate_from_embeds(x, max_length=L).clone()
            t0 = time.time(); n = 0
            while time.time() - t0 < 6.0:
                out = m.generate_from_embeds(x, max_length=L)
                n += 1
                if out.shape != ref.shape or not torch.equal(out, ref):
                    bad += 1
            print(f"{prec} eos={eos} B={B} S={S} L={L}: {n} repeats, mismatches so far {bad}", flush=True)
print("SOAK", "FAILED" if bad else "OK")
