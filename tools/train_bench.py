#!/usr/bin/env python3
"""Time the native training step (BASELINE configs[4] per-GPU shape: batch 128 / 8 GPUs = 16 clips)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from music2midi_amd import synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import T5Geometry, default_config
from music2midi_amd.training import NativeTrainer
from music2midi_amd.transformer import T5Transformer

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"  # bf16 | fp32 | fp8
cfg = default_config(); geom = T5Geometry(cfg.model.t5)
sd = synth.t5_state_dict(geom, 0)
model = T5Transformer(cfg.to_dict(), precision="fp32"); load_t5_state(model, sd, strict=False); model = model.cuda()
for (B, S, Ld) in ((16, 190, 128), (16, 261, 256), (64, 261, 256)):
    tr = NativeTrainer(model, B, S, Ld, precision=prec)
    if "dropout" in sys.argv[2:]:                          # T5Config.dropout_rate = 0.1, as the reference trains (train() mode)
        tr.set_dropout(0.1, 1)
    if len(sys.argv) > 2 and sys.argv[2] == "split":       # the two-part backward pass of the data-parallel overlap (one GPU: what the split costs)
        tr.set_sync_stream(torch.cuda.Stream())
    x = torch.from_numpy(synth.normal(1, "x", (B, S, 384), 2.0)).cuda()
    cond = torch.from_numpy(synth.cond_index_batch(0, B)).cuda()
    labels = (torch.from_numpy((synth.uniform01(4, "l", B * Ld) * 330).astype(np.int64).reshape(B, Ld)) + 3).cuda()
    for _ in range(2): tr.forward_backward(x, cond, labels); tr.optimizer_step()
    torch.cuda.synchronize(); n = 10
    t0 = time.perf_counter()
    for _ in range(n): tr.forward_backward(x, cond, labels, backward=False)
    torch.cuda.synchronize(); t_f = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n): tr.forward_backward(x, cond, labels)
    torch.cuda.synchronize(); t_fb = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n): tr.optimizer_step()
    torch.cuda.synchronize(); t_o = (time.perf_counter() - t0) / n
    enc = 6 * (4227072 * S + 2048 * S * S) + 4718592 * S
    dec = Ld * (6 * (2 * 384 * 512 * 6 + 3 * 2 * 384 * 1152) + 2 * 384 * 400) + 6 * (2 * 2 * Ld * Ld * 512 + 2 * 2 * Ld * S * 512)
    fl = B * (enc + dec)
    print(f"{prec} B={B} S={S} Ld={Ld}: forward {t_f*1e3:.2f} ms, fwd+bwd {t_fb*1e3:.2f} ms ({3*fl/t_fb/1e12:.1f} TFLOP/s), adafactor {t_o*1e3:.2f} ms, "
          f"step {1e3*(t_fb+t_o):.2f} ms = {B*Ld/(t_fb+t_o):.0f} label tokens/s, {B/(t_fb+t_o):.1f} clips/s", flush=True)
    tr.close()
