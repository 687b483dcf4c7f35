#!/usr/bin/env python3
"""Print digests of bf16-mode results whose every projection goes through gemm_kernel: encoder states + greedy ids at two
geometries, and loss + gradients of training steps at three shapes.  tests/test_gemm_dma_gpu.py runs this once per setting of
M2M_GEMM_DMA / M2M_GEMM_DMA_NS1 / _NS2 (read once per process) and compares the digests: the LDS-DMA main loop accumulates every
output element in the same k order as the register-staged one, so the results must be bit-identical."""
import copy, hashlib, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
import numpy as np, torch
from music2midi_amd import synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import DEFAULT_CONFIG, T5Geometry, load_config
from music2midi_amd.training import NativeTrainer
from music2midi_amd.transformer import T5Transformer
from test_t5_gpu import tiny_config


def digest(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]


for name, cfg, shapes_inf, shapes_tr in (("tiny", tiny_config(), [(3, 19, 12)], [(3, 21, 14), (2, 70, 33)]),
                                         ("full", copy.deepcopy(DEFAULT_CONFIG), [(3, 190, 24), (2, 864, 16), (32, 864, 4)],
                                          [(4, 188, 48), (16, 259, 256)])):
    geom = T5Geometry(load_config(cfg).model.t5)
    sd = synth.t5_state_dict(geom, seed=0)
    synth.perturb_layer_norms(sd, 0)
    m = T5Transformer(cfg, precision="bf16")
    load_t5_state(m, sd, strict=False)
    m = m.cuda().eval()
    for B, S, L in shapes_inf:
        x = torch.from_numpy(synth.normal(7, "embeds", (B, S, geom.d_model), 3.0)).cuda()
        print(name, "encode", B, S, digest(m.encode(x)), "ids", digest(m.generate_from_embeds(x, max_length=L)))
    mt = T5Transformer(cfg, precision="fp32")
    load_t5_state(mt, sd, strict=False)
    mt = mt.cuda()
    for B, F, Ld in shapes_tr:
        tr = NativeTrainer(mt, B, F + 2, Ld, precision="bf16")
        tr.set_dropout(0.1, seed=9)
        x = torch.zeros((B, F + 2, geom.d_model))
        x[:, 2:] = torch.from_numpy(synth.normal(5, "feats", (B, F, geom.d_model), 2.0))
        cond = torch.from_numpy(synth.cond_index_batch(2, B))
        labels = torch.from_numpy((synth.uniform01(4, "labels", B * Ld) * 330).astype(np.int64).reshape(B, Ld)) + 3
        loss, _ = tr.forward_backward(x.cuda(), cond.cuda(), labels.cuda())
        print(name, "train", B, F, Ld, "loss", repr(loss.item()), "grads", digest(tr.grads))
        tr.close()
