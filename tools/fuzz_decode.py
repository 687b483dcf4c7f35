#!/usr/bin/env python3
"""Differential fuzz of the greedy decode loop against the CPU oracle (fp32): random batch sizes, encoder lengths,
max_length, EOS-prone heads, chain splits, graph lengths.  Ids must be identical; the only excuse is a row whose FIRST
difference sits on a step where the oracle's own top-2 logit margin is below NEAR_TIE (fp32 summation-order noise: a few
ulp of the logit) -- such rows are checked up to that step and counted separately.
python tools/fuzz_decode.py [cases] [seed]"""
import copy, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from music2midi_amd import synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import DEFAULT_CONFIG, T5Geometry, load_config
from music2midi_amd.transformer import T5Transformer
from oracle.t5 import T5Oracle

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
torch.set_num_threads(16)
tiny = copy.deepcopy(DEFAULT_CONFIG); tiny["model"]["t5"].update(d_model=128, d_ff=256, num_layers=2, num_decoder_layers=2, num_heads=2)
models = {}
# kernel forms of the decode step (round 6): (clips per attention workgroup, hidden slices per feed-forward workgroup), latched when a
# model's session is created - 0 = by chain size.  Every form must give the oracle's ids at every shape, however small the chain.
FORMS = [(0, 0), (2, 2), (4, 4), (4, 1), (2, 4), (1, 1)]
def get(cfg_name, eos_kind, form=(0, 0)):
    key = (cfg_name, eos_kind, form)
    os.environ["M2M_DA_CLIPS"], os.environ["M2M_DEC_FF_SLICES"] = str(form[0]), str(form[1])     # read by the session this model creates on first use
    if key not in models:
        cfg = tiny if cfg_name == "tiny" else DEFAULT_CONFIG
        g = T5Geometry(load_config(cfg).model.t5)
        sd = synth.t5_state_dict(g, seed=1); synth.perturb_layer_norms(sd, 1)
        if eos_kind == 1: synth.force_eos_head(sd, g)
        if eos_kind == 2: synth.force_eos_head(sd, g, active=340, eos_scale=1.6)
        m = T5Transformer(cfg, precision="fp32"); load_t5_state(m, sd, strict=False)
        models[key] = (m.cuda().eval(), T5Oracle(g, sd), g)
    return models[key]
u = synth.uniform01(seed, "fuzz", n_cases * 8).reshape(n_cases, 8)
NEAR_TIE = 1e-4
bad = 0; near = 0; t0 = time.time()
for i, r in enumerate(u):
    cfg_name = "tiny" if r[0] < 0.6 else "full"
    eos_kind = int(r[1] * 3)
    B = 1 + int(r[2] * (40 if cfg_name == "tiny" else 12))
    S = 3 + int(r[3] * (300 if cfg_name == "tiny" else 120))
    L = 1 + int(r[4] * (80 if cfg_name == "tiny" else 40))
    rows = [0, 1, 3, 7, 16][int(r[5] * 5)]
    os.environ.pop("M2M_GROUP_ROWS", None)
    if rows: os.environ["M2M_GROUP_ROWS"] = str(rows)
    os.environ["M2M_GRAPH_STEPS"] = str([1, 3, 8][int(r[6] * 3)])
    form = FORMS[int(r[7] * len(FORMS))]
    m, orc, g = get(cfg_name, eos_kind, form)
    x = torch.from_numpy(synth.normal(1000 + i, "x", (B, S, g.d_model), 3.0))
    want, margins = orc.generate(x, L, return_margins=True)
    got = m.generate_from_embeds(x.cuda(), max_length=L).cpu()
    ok = torch.equal(got, want)
    if not ok:
        # per row: everything before the first difference is identical by construction; the difference itself must be a near-tie.
        # (a near-tie flip can change WHEN the last row finishes, hence the returned length: compare the common prefix)
        n = min(got.shape[1], want.shape[1])
        excused = got.shape[0] == want.shape[0]
        worst = 0.0
        for b in range(want.shape[0] if excused else 0):
            d = (got[b, :n] != want[b, :n]).nonzero()
            if len(d):
                t = int(d[0, 0]); mg = float(margins[b, t - 1]); worst = max(worst, mg)
                excused &= mg < NEAR_TIE
        if excused and (worst > 0 or got.shape == want.shape):
            near += 1
            print(f"near-tie case {i}: {cfg_name} eos={eos_kind} B={B} S={S} L={L} rows={rows}: first differences at oracle margin <= {worst:.2e}", flush=True)
            continue
        bad += 1
        print(f"MISMATCH case {i}: {cfg_name} eos={eos_kind} form={form} B={B} S={S} L={L} rows={rows} graph={os.environ['M2M_GRAPH_STEPS']} got {tuple(got.shape)} want {tuple(want.shape)}", flush=True)
    elif i % 10 == 0:
        print(f"case {i}: {cfg_name} eos={eos_kind} form={form} B={B} S={S} L={L} rows={rows} -> ids {tuple(got.shape)} identical ({time.time() - t0:.0f} s)", flush=True)
print("FUZZ", "FAILED" if bad else "OK", f"{n_cases} cases, {bad} mismatches, {near} excused near-ties (oracle top-2 margin < {NEAR_TIE:g})")
sys.exit(1 if bad else 0)
