#!/usr/bin/env python3
"""Layer-count bisect: teacher-forced logits fused vs two-kernel for (enc layers, dec layers) combinations; fused run twice."""
import copy, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from music2midi_amd import synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import DEFAULT_CONFIG, T5Geometry, load_config
from music2midi_amd.transformer import T5Transformer
B, S, Ld = (int(v) for v in sys.argv[1:4])
for ne, nd in ((1, 1), (2, 1), (6, 1), (1, 2), (1, 6), (6, 6)):
    cfgd = copy.deepcopy(DEFAULT_CONFIG)
    cfgd["model"]["t5"].update(num_layers=ne, num_decoder_layers=nd)
    geom = T5Geometry(load_config(cfgd).model.t5)
    sd = synth.t5_state_dict(geom, seed=0); synth.perturb_layer_norms(sd, 0)
    m = T5Transformer(cfgd, precision="bf16"); load_t5_state(m, sd, strict=False); m = m.cuda().eval()
    x = torch.from_numpy(synth.normal(7, "embeds", (B, S, geom.d_model), 3.0)).cuda()
    dec = torch.from_numpy((synth.uniform01(11, "dec", B * Ld) * (geom.vocab_size - 3)).astype(np.int64).reshape(B, Ld) + 3).cuda()
    dec[:, 0] = geom.decoder_start_token_id
    res = {}
    for flag in ("0", "0", "1", "1", "e3", "e5"):
        os.environ["M2M_NORM_GEMM"] = flag
        enc = m.encode(x).cpu()
        lg = m.logits_from_embeds(x, dec).cpu()
        res.setdefault(flag, []).append((enc, lg))
    ref_enc, ref = res["0"][0]
    msg = [f"enc {ne} dec {nd}: old twice same {torch.equal(ref, res['0'][1][1])}; fused twice same {torch.equal(res['1'][0][1], res['1'][1][1])}"]
    for flag in ("1", "e3", "e5"):
        enc, lg = res[flag][0]
        msg.append(f"{flag}: enc states differ {int((enc != ref_enc).sum())}, logits differ {int((lg != ref).sum())}")
    print("; ".join(msg), flush=True)
