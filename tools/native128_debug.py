"""Diagnostic: B = 128 x S = 190 greedy decode of tiled copies of the two native_s190 fixture clips: where do rows differ?"""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from forced_check import case_inputs
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import DEFAULT_CONFIG, T5Geometry
from music2midi_amd.transformer import T5Transformer
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
z = np.load(ROOT / "tests/golden/t5_forced.npz")
g = T5Geometry(DEFAULT_CONFIG["model"]["t5"])
sd, x = case_inputs(f"native_s190_{prec}", g, "cuda")
want = z[f"native_s190_{prec}/ids"].astype(np.int64)
for copies in (1, 8, 16, 32, 33, 48, 64):
    m = T5Transformer(DEFAULT_CONFIG, precision=prec)
    load_t5_state(m, sd, strict=False)
    m = m.cuda().eval()
    ids = m.generate_from_embeds(x.repeat(copies, 1, 1).contiguous(), max_length=1024).cpu().numpy()
    L = ids.shape[1]
    bad_rows = [r for r in range(ids.shape[0]) if not np.array_equal(ids[r], want[r % 2, :L])]
    first = {r: int(np.nonzero(ids[r] != want[r % 2, :L])[0][0]) for r in bad_rows[:8]}
    print(f"copies {copies:3d} (B={2 * copies}): out len {L}, rows differing from the fixture: {len(bad_rows)} {bad_rows[:12]} first diff step {first}", flush=True)
    del m
