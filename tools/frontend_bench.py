#!/usr/bin/env python3
"""Time the fused log-mel kernel alone (BASELINE configs[1]: 64 clips x 220 500 samples)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from music2midi_amd import synth
from music2midi_amd.input import LogMelSpectrogram

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = 220500
x = torch.from_numpy(synth.waveform_batch(0, B, T)).cuda()
fe = LogMelSpectrogram(16000, 2048, 256, 20.0, 384)
out = torch.empty(B, 864, 384, device="cuda")
for _ in range(3): fe.forward_into(x, out, 2)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 20
e0.record()
for _ in range(N): fe.forward_into(x, out, 2)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1000 / N
bytes_alg = B * (4 * T + 4 * 862 * 384)
print(f"B={B}: {us:.1f} us per launch, {bytes_alg / us / 1e3:.1f} GB/s algorithmic ({bytes_alg / us / 1e3 / 8000:.3%} of 8 TB/s), "
      f"{B * 56.8e6 / us / 1e6:.2f} TFLOP/s")
