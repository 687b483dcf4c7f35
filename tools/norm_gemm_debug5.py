#!/usr/bin/env python3
"""Two encoder layers: workspace after m2m_encode, fused (panel also written to h_enc) vs two-kernel path."""
import copy, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from music2midi_amd import synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import DEFAULT_CONFIG, T5Geometry, load_config
from music2midi_amd.transformer import T5Transformer
B, S, L = int(sys.argv[1]), int(sys.argv[2]), 8
NE = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cfgd = copy.deepcopy(DEFAULT_CONFIG)
cfgd["model"]["t5"].update(num_layers=NE, num_decoder_layers=1)
geom = T5Geometry(load_config(cfgd).model.t5)
sd = synth.t5_state_dict(geom, seed=0); synth.perturb_layer_norms(sd, 0)
m = T5Transformer(cfgd, precision="bf16"); load_t5_state(m, sd, strict=False); m = m.cuda().eval()
x = torch.from_numpy(synth.normal(7, "embeds", (B, S, geom.d_model), 3.0)).cuda()
d, inner, dff, es = geom.d_model, geom.num_heads * geom.d_kv, geom.d_ff, 2
al = lambda v: (v + 255) // 256 * 256
Ma = B * max(S, L); Spa = (max(S, L) + 63) // 64 * 64; M = B * S
sizes = [("x_enc", Ma * d * 4), ("h_enc", Ma * d * es), ("qkv_enc", 3 * Ma * inner * es), ("vt_enc", B * inner * Spa * es), ("attn_enc", Ma * inner * es), ("mid_enc", Ma * dff * es)]
os.environ["M2M_NORM_GEMM_HOUT"] = "1"
snaps = {}
for flag in ("1", "0"):
    os.environ["M2M_NORM_GEMM"] = flag
    m._encode(x, L); torch.cuda.synchronize()
    base = (m._workspace.data_ptr() + 255) // 256 * 256 - m._workspace.data_ptr()
    snaps[flag] = m._workspace[base:].clone().cpu().numpy()
off = 0
def bf(a): return (a.astype(np.uint32) << 16).view(np.float32)
for name, nb in sizes:
    a, b = snaps["1"][off:off + nb], snaps["0"][off:off + nb]; off = al(off + nb)
    dt = np.float32 if name == "x_enc" else np.uint16
    av, bv = a.view(dt), b.view(dt)
    if name == "x_enc": av, bv = av[: M * d].reshape(M, d), bv[: M * d].reshape(M, d)
    elif name == "h_enc": av, bv = av[: M * d].reshape(M, d), bv[: M * d].reshape(M, d)
    elif name == "qkv_enc": av, bv = av[: 2 * Ma * inner].reshape(2, B, 8, S, 64), bv[: 2 * Ma * inner].reshape(2, B, 8, S, 64)
    elif name == "vt_enc": av, bv = av.reshape(B, 8, 64, Spa)[..., :S], bv.reshape(B, 8, 64, Spa)[..., :S]
    elif name == "attn_enc": av, bv = av[: M * inner].reshape(M, inner), bv[: M * inner].reshape(M, inner)
    else: av, bv = av[: M * dff].reshape(M, dff), bv[: M * dff].reshape(M, dff)
    ne = np.argwhere(av != bv)
    print(f"{name:9s}: differing {len(ne)} of {av.size}", ne[:8].tolist())
    if len(ne) and name != "x_enc":
        i = tuple(ne[0]); print("     values fused / old:", bf(av)[i], bf(bv)[i])
    if len(ne) and name == "x_enc":
        i = tuple(ne[0]); print("     values fused / old:", av[i], bv[i])

