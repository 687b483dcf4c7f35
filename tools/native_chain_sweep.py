"""Reference-native geometry (128 segments x 48 000 samples, S = 190, 1 024 tokens, bf16): tokens/s by clips per decode chain
(M2M_GROUP_ROWS; a child process per setting — the variable is read when the session plans its chains)."""
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
m = T5Transformer(DEFAULT_CONFIG, precision="bf16"); load_t5_state(m, synth.t5_state_dict(g, seed=0), strict=False); m = m.cuda().eval()
B = int(sys.argv[1]); T = int(sys.argv[2])
wav = torch.from_numpy(synth.waveform_batch(1000, B, T)).cuda(); cond = torch.from_numpy(synth.cond_index_batch(1000, B)).cuda()
inp = ModelInputs(input_waveform=wav, cond_index=cond)
t = m.generate(inp, max_length=1024); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): t = m.generate(inp, max_length=1024)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"{B * (t.shape[1] - 1) / dt / 1e3:.1f} k tok/s, {dt * 1e3:.1f} ms per batch, {dt / (t.shape[1] - 1) * 1e6:.1f} us per step")
''' % str(ROOT)
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 48000)
for rows in sys.argv[3:] or ["64", "43", "32", "22", "16"]:
    env = dict(os.environ, M2M_GROUP_ROWS=rows)
    r = subprocess.run([sys.executable, "-c", CODE, str(B), str(T)], env=env, capture_output=True, text=True, timeout=900)
    print(f"B={B} T={T} M2M_GROUP_ROWS={rows}: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
