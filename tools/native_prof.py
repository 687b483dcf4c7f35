"""Reference-native geometry under rocprofv3: 128 x 48 000 samples, S = 190, 256 tokens (kernel-stats run)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from music2midi_amd import synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import DEFAULT_CONFIG, T5Geometry
from music2midi_amd.input import ModelInputs
from music2midi_amd.transformer import T5Transformer
g = T5Geometry(DEFAULT_CONFIG["model"]["t5"])
m = T5Transformer(DEFAULT_CONFIG, precision="bf16"); load_t5_state(m, synth.t5_state_dict(g, seed=0), strict=False); m = m.cuda().eval()
B, T, L = int(sys.argv[1]) if len(sys.argv) > 1 else 128, 48000, int(sys.argv[2]) if len(sys.argv) > 2 else 1024
wav = torch.from_numpy(synth.waveform_batch(1000, B, T)).cuda(); cond = torch.from_numpy(synth.cond_index_batch(1000, B)).cuda()
for _ in range(2): t = m.generate(ModelInputs(input_waveform=wav, cond_index=cond), max_length=L)
torch.cuda.synchronize(); print(t.shape)
