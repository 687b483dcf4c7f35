#!/usr/bin/env python3
"""Hash of loss + flat gradient after one training pass (bf16 / fp8 / fp32, dropout on) — for checking that a switch which only
moves work between launches leaves every bit alone:  M2M_...=0 python tools/train_grad_hash.py bf16  vs  the default."""
import hashlib, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from music2midi_amd import synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import T5Geometry, default_config
from music2midi_amd.training import NativeTrainer
from music2midi_amd.transformer import T5Transformer

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
cfg = default_config(); geom = T5Geometry(cfg.model.t5)
model = T5Transformer(cfg.to_dict(), precision="fp32"); load_t5_state(model, synth.t5_state_dict(geom, 0), strict=False); model = model.cuda()
for (B, S, Ld) in ((3, 77, 40), (16, 261, 256)):
    tr = NativeTrainer(model, B, S, Ld, precision=prec)
    tr.set_dropout(0.1, 5)
    x = torch.from_numpy(synth.normal(1, "x", (B, S, 384), 2.0)).cuda(); cond = torch.from_numpy(synth.cond_index_batch(0, B)).cuda()
    labels = (torch.from_numpy((synth.uniform01(4, "l", B * Ld) * 330).astype(np.int64).reshape(B, Ld)) + 3).cuda()
    for it in range(3):
        loss, _ = tr.forward_backward(x, cond, labels)
        torch.cuda.synchronize()
        h = hashlib.sha256(tr.grads.cpu().numpy().tobytes()).hexdigest()[:16]
        print(f"{prec} B={B} S={S} Ld={Ld} call {it}: loss {loss.item():.9f} grads {h}", flush=True)
    tr.close()
