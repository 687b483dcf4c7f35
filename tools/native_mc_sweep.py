"""Large-chain decode forms (round 6): tokens/s by clips per attention workgroup (M2M_DA_CLIPS), rows per feed-forward workgroup
(M2M_DEC_FF_ROWS) and clips per chain (M2M_GROUP_ROWS), a child process per setting (the switches are latched per session).

    python tools/native_mc_sweep.py B T precision max_length  "clips,ffrows,grouprows[,ffslices]" ...
"""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
CODE = r'''
import sys, time, torch
sys.path.insert(0, %r)
from music2midi_amd import synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import DEFAULT_CONFIG, T5Geometry
from music2midi_amd.input import ModelInputs
from music2midi_amd.transformer import T5Transformer
g = T5Geometry(DEFAULT_CONFIG["model"]["t5"])
B = int(sys.argv[1]); T = int(sys.argv[2]); prec = sys.argv[3]; L = int(sys.argv[4])
m = T5Transformer(DEFAULT_CONFIG, precision=prec); load_t5_state(m, synth.t5_state_dict(g, seed=0), strict=False); m = m.cuda().eval()
wav = torch.from_numpy(synth.waveform_batch(1000, B, T)).cuda(); cond = torch.from_numpy(synth.cond_index_batch(1000, B)).cuda()
inp = ModelInputs(input_waveform=wav, cond_index=cond)
t = m.generate(inp, max_length=L); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): t = m.generate(inp, max_length=L)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"{B * (t.shape[1] - 1) / dt / 1e3:.1f} k tok/s, {dt * 1e3:.1f} ms per batch, {dt / (t.shape[1] - 1) * 1e6:.1f} us per step, checksum {int(t.sum())}")
''' % str(ROOT)
B, T, prec, L = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
for spec in sys.argv[5:]:
    clips, ffr, rows, *rest = spec.split(",")
    env = dict(os.environ, M2M_DA_CLIPS=clips, M2M_DEC_FF_ROWS=ffr)
    if rest:
        env["M2M_DEC_FF_SLICES"] = rest[0]
    if rows != "0":
        env["M2M_GROUP_ROWS"] = rows
    r = subprocess.run([sys.executable, "-c", CODE, str(B), str(T), prec, str(L)], env=env, capture_output=True, text=True, timeout=900)
    print(f"B={B} T={T} {prec} L={L} clips/wg={clips} ff_rows={ffr} group_rows={rows} ff_slices={rest[0] if rest else 'auto'}: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]}", flush=True)
