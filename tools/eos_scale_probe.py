"""Diagnostic: EOS positions of a batch of synthetic encoder inputs under synth.force_eos_head for a few eos_scale values (tuning bench.py's ragged_eos)."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from music2midi_amd import synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import DEFAULT_CONFIG, T5Geometry
from music2midi_amd.input import ModelInputs
from music2midi_amd.transformer import T5Transformer
g = T5Geometry(DEFAULT_CONFIG["model"]["t5"])
B, S = int(sys.argv[1]), int(sys.argv[2])
x = torch.from_numpy(synth.normal(21, "embeds", (B, S, g.d_model), 3.0)).cuda()
for perturb in (1,):
    for active, scale in ((340, 1.6), (340, 1.4), (340, 1.3), (340, 1.2), (340, 1.1), (340, 1.0), (340, 0.9)):
        sd = synth.t5_state_dict(g, seed=0)
        if perturb: synth.perturb_layer_norms(sd, 0)
        synth.force_eos_head(sd, g, active=active, eos_scale=scale)
        m = T5Transformer(DEFAULT_CONFIG, precision="bf16"); load_t5_state(m, sd, strict=False); m = m.cuda().eval()
        a = m.generate_from_embeds(x, max_length=1024).cpu().numpy()
        ends = sorted(int(np.nonzero(a[r] == 2)[0][0]) if (a[r] == 2).any() else 9999 for r in range(B))
        print(f"perturb {perturb} active {active} scale {scale}: len {a.shape[1]} ends q0/q25/q50/q75/q100 {[ends[int(q * (B - 1))] for q in (0, .25, .5, .75, 1)]} no-eos rows {sum(e == 9999 for e in ends)}", flush=True)
        del m
