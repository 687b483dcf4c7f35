"""GPU parity: HIP T5 encoder / teacher-forced logits / greedy ids vs the CPU oracle."""
import copy

import numpy as np
import pytest
import torch

from music2midi_amd import synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import DEFAULT_CONFIG, T5Geometry, load_config
from music2midi_amd.transformer import T5Transformer

pytestmark = pytest.mark.gpu


def tiny_config():
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    cfg["model"]["t5"].update(d_model=128, d_ff=256, num_layers=2, num_decoder_layers=2, num_heads=2)
    return cfg


def build(cfg_dict, precision, seed=0, eos=False):
    cfg = load_config(cfg_dict)
    geom = T5Geometry(cfg.model.t5)
    sd = synth.t5_state_dict(geom, seed=seed)
    synth.perturb_layer_norms(sd, seed)
    if eos:
        synth.force_eos_head(sd, geom)
    model = T5Transformer(cfg_dict, precision=precision)
    load_t5_state(model, sd, strict=False)
    model = model.cuda().eval()
    from oracle.t5 import T5Oracle
    return model, T5Oracle(geom, sd, emulate=precision), geom


def embeds(B, S, d, seed=7):
    return torch.from_numpy(synth.normal(seed, "embeds", (B, S, d), 3.0))


@pytest.mark.parametrize("cfg_name,B,S", [("tiny", 3, 19), ("tiny", 2, 130), ("full", 2, 190), ("full", 33, 40), ("full", 1, 1100)])
def test_encoder_fp32(cfg_name, B, S):      # S = 1100: a relative-position table of more than 2 048 entries (staged in two passes)
    cfg = tiny_config() if cfg_name == "tiny" else DEFAULT_CONFIG
    model, orc, g = build(cfg, "fp32")
    x = embeds(B, S, g.d_model)
    ref = orc.encode(x)
    out = model.encode(x.cuda()).cpu()
    err = (out - ref).abs().max().item()
    print(f"encoder fp32 {cfg_name} B={B} S={S}: max|diff|={err:.3e} (ref absmax {ref.abs().max():.2f})")
    assert err < 2e-4


@pytest.mark.parametrize("mode", ["batched", "step"])
@pytest.mark.parametrize("cfg_name,B,S,Ld", [("tiny", 3, 19, 12), ("full", 2, 190, 24), ("full", 5, 64, 40), ("full", 3, 70, 200),
                                             ("tiny", 2, 9, 1), ("tiny", 9, 130, 131), ("full", 17, 21, 129)])
def test_forced_logits_fp32(monkeypatch, mode, cfg_name, B, S, Ld):
    """Teacher-forced logits: the batched pass (MFMA GEMMs + causal flash attention over all positions) and the
    KV-cached decode steps in forced mode must both reproduce the oracle."""
    monkeypatch.setenv("M2M_FORWARD", mode)
    cfg = tiny_config() if cfg_name == "tiny" else DEFAULT_CONFIG
    model, orc, g = build(cfg, "fp32")
    x = embeds(B, S, g.d_model)
    labels = torch.from_numpy((synth.uniform01(3, "labels", B * Ld) * 330).astype(np.int64).reshape(B, Ld)) + 3
    _, ref = orc.forward(x, labels)
    dec_in = torch.full_like(labels, g.decoder_start_token_id)
    dec_in[:, 1:] = labels[:, :-1]
    out = model.logits_from_embeds(x.cuda(), dec_in.cuda()).cpu()
    err = (out - ref).abs().max().item()
    print(f"forced logits fp32 {cfg_name}: max|diff|={err:.3e} (ref absmax {ref.abs().max():.1f})")
    assert err < 2e-3


@pytest.mark.parametrize("cfg_name,B,S,L,eos", [("tiny", 3, 19, 40, False), ("full", 2, 190, 64, False),
                                                 ("full", 4, 60, 96, True), ("tiny", 5, 30, 64, True)])
def test_greedy_ids_fp32_bit_exact(cfg_name, B, S, L, eos):
    cfg = tiny_config() if cfg_name == "tiny" else DEFAULT_CONFIG
    model, orc, g = build(cfg, "fp32", eos=eos)
    x = embeds(B, S, g.d_model)
    ref, margins = orc.generate(x, L, return_margins=True)
    out = model.generate_from_embeds(x.cuda(), max_length=L).cpu()
    print(f"greedy fp32 {cfg_name} eos={eos}: ref shape {tuple(ref.shape)} out {tuple(out.shape)} "
          f"min margin {margins.min().item():.4f}")
    print(ref[:, :24])
    assert out.shape == ref.shape
    assert torch.equal(out, ref)


@pytest.mark.parametrize("cfg_name,B,S,L", [("tiny", 3, 19, 40), ("full", 2, 190, 64), ("full", 1, 1100, 6), ("full", 2, 96, 12)])
def test_bf16_mode_tracks_bf16_oracle(cfg_name, B, S, L):      # S = 1100: table > 2 048 entries; S = 96: one wide step + the half step
    cfg = tiny_config() if cfg_name == "tiny" else DEFAULT_CONFIG
    model, orc, g = build(cfg, "bf16")
    x = embeds(B, S, g.d_model)
    ref_enc = orc.encode(x)
    out_enc = model.encode(x.cuda()).cpu()
    e = (out_enc - ref_enc).abs().max().item()
    print(f"encoder bf16 {cfg_name}: max|diff|={e:.3e} rel-to-absmax {e / ref_enc.abs().max().item():.3e}")
    assert e < 0.08
    ref, margins = orc.generate(x, L, return_margins=True)
    out = model.generate_from_embeds(x.cuda(), max_length=L).cpu()
    n = min(out.shape[1], ref.shape[1])
    agree = (out[:, :n] == ref[:, :n]).float().mean().item()
    print(f"greedy bf16 {cfg_name}: agreement {agree:.3f}, min margin {margins.min().item():.4f}")
    # exact wherever the oracle's top-2 margin is comfortably above bf16 noise, up to the first divergence
    for b in range(B):
        for t in range(1, n):
            if out[b, t] != ref[b, t]:
                assert margins[b, t - 1] < 0.5, f"row {b} step {t}: diverged at margin {margins[b, t-1]:.3f}"
                break


@pytest.mark.parametrize("B,S,L,rows", [(1, 7, 20, 32), (17, 11, 24, 32), (33, 9, 16, 32), (40, 5, 12, 16), (9, 300, 20, 4),
                                        (140, 5, 10, 140)])   # 280 attention workgroups > 256 CUs: late starters
def test_ragged_batches_and_decode_chains_fp32(monkeypatch, B, S, L, rows):
    """Batch sizes that are not multiples of the 16-row MFMA tile, more clips than one chain holds,
    several chains of unequal size (M2M_GROUP_ROWS), S beyond one key round: ids must not change."""
    monkeypatch.setenv("M2M_GROUP_ROWS", str(rows))
    model, orc, g = build(tiny_config(), "fp32", eos=(B % 2 == 1))
    x = embeds(B, S, g.d_model, seed=B)
    ref = orc.generate(x, L)
    out = model.generate_from_embeds(x.cuda(), max_length=L).cpu()
    assert out.shape == ref.shape and torch.equal(out, ref)


def test_max_length_edge_cases_and_session_reuse():
    model, orc, g = build(tiny_config(), "fp32")
    x = embeds(3, 10, g.d_model)
    for L in (1, 2, 3, 17):                       # max_length 1 -> only the start token
        out = model.generate_from_embeds(x.cuda(), max_length=L).cpu()
        assert torch.equal(out, orc.generate(x, L)), L
    # a smaller problem after a bigger one re-uses the session; a bigger one re-creates it
    big = embeds(6, 40, g.d_model, seed=2)
    assert torch.equal(model.generate_from_embeds(big.cuda(), max_length=30).cpu(), orc.generate(big, 30))
    assert torch.equal(model.generate_from_embeds(x.cuda(), max_length=9).cpu(), orc.generate(x, 9))


def test_no_graph_mode_matches_graph_mode(monkeypatch):
    model, orc, g = build(tiny_config(), "bf16")
    x = embeds(4, 21, g.d_model).cuda()
    a = model.generate_from_embeds(x, max_length=40)
    monkeypatch.setenv("M2M_NO_GRAPH", "1")
    b = model.generate_from_embeds(x, max_length=40)
    assert torch.equal(a, b)


def test_error_paths_report_instead_of_faulting():
    from music2midi_amd import native
    model, _, g = build(tiny_config(), "fp32")
    with pytest.raises(native.NativeError, match="too short"):
        model.spectrogram(torch.zeros(1, 512, device="cuda"))
    with pytest.raises(native.NativeError):
        model.generate_from_embeds(torch.zeros(1, 40000, g.d_model, device="cuda"), max_length=4)   # S beyond the LDS bias table


@pytest.mark.parametrize("d_model,d_ff,heads,layers", [(256, 512, 4, 1), (512, 1152, 2, 1), (384, 256, 8, 2), (128, 640, 6, 1)])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_other_geometries(d_model, d_ff, heads, layers, precision):
    """Every d_model the decode kernels are instantiated for (128/256/384/512), feed-forward widths other than
    the reference's (any multiple of 64) and head counts other than 8, in both precision modes."""
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    cfg["model"]["t5"].update(d_model=d_model, d_ff=d_ff, num_layers=layers, num_decoder_layers=layers, num_heads=heads)
    model, orc, g = build(cfg, precision)
    x = embeds(3, 37, g.d_model)
    ref_enc = orc.encode(x)
    out_enc = model.encode(x.cuda()).cpu()
    tol = 3e-4 if precision == "fp32" else 0.1
    assert (out_enc - ref_enc).abs().max().item() < tol
    ref, margins = orc.generate(x, 24, return_margins=True)
    out = model.generate_from_embeds(x.cuda(), max_length=24).cpu()
    if precision == "fp32":
        assert torch.equal(out, ref)
    else:
        n = min(out.shape[1], ref.shape[1])
        for b in range(3):
            for t in range(1, n):
                if out[b, t] != ref[b, t]:
                    assert margins[b, t - 1] < 0.5
                    break


# ------------------------------------------------------------------ fixed-point residual range guard
def _generate_with(sd_edit, precision="fp32"):
    from music2midi_amd import native
    cfg = tiny_config()
    geom = T5Geometry(load_config(cfg).model.t5)
    sd = synth.t5_state_dict(geom, seed=0)
    sd_edit(sd)
    model = T5Transformer(cfg, precision=precision)
    load_t5_state(model, sd, strict=False)
    model = model.cuda().eval()
    x = embeds(3, 19, geom.d_model).cuda()
    return model, x, native


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_out_of_range_residual_is_an_error_not_silent_garbage(precision):
    """The decoder's residual stream is int64 fixed point, valid for |x| < 2^21.  Weights that push it past that
    (or Inf/NaN weights) must surface as M2M_ERR_RANGE (-5) — the fp32 reference would give Inf/NaN logits —
    and the session must stay usable afterwards."""
    def huge_embedding(sd):
        sd["transformer.shared.weight"] = sd["transformer.shared.weight"] * 4.0e6        # |x| ~ 4e6 > 2^21
    model, x, native = _generate_with(huge_embedding, precision)
    with pytest.raises(native.NativeError, match=r"status -5.*fixed-point residual range"):
        model.generate_from_embeds(x, max_length=12)

    def inf_in_a_decoder_weight(sd):
        sd["transformer.decoder.block.1.layer.2.DenseReluDense.wo.weight"][7, 5] = np.inf
    model, x, native = _generate_with(inf_in_a_decoder_weight, precision)
    with pytest.raises(native.NativeError, match="status -5"):
        model.generate_from_embeds(x, max_length=12)

    def nan_in_the_head(sd):
        sd["transformer.lm_head.weight"][11, 3] = np.nan
    model, x, native = _generate_with(nan_in_the_head, precision)
    with pytest.raises(native.NativeError, match="status -5"):
        model.generate_from_embeds(x, max_length=12)

    # large but legal activations (|x| ~ 1e5) are fine, and a clean run after a failed one works on the same process
    def big_but_legal(sd):
        sd["transformer.shared.weight"] = sd["transformer.shared.weight"] * 3.0e4
    model, x, native = _generate_with(big_but_legal, precision)
    ids = model.generate_from_embeds(x, max_length=12)
    assert ids.shape == (3, 12) and torch.equal(ids, model.generate_from_embeds(x, max_length=12))


def test_large_activation_fp32_still_matches_oracle():
    """Residual values up to ~1e5 (2^-30 resolution, 2^21 range): ids still bit-equal to the fp32 oracle."""
    cfg = tiny_config()
    geom = T5Geometry(load_config(cfg).model.t5)
    sd = synth.t5_state_dict(geom, seed=0)
    synth.perturb_layer_norms(sd, 0)
    sd["transformer.shared.weight"] = sd["transformer.shared.weight"] * 2.0e3
    model = T5Transformer(cfg, precision="fp32")
    load_t5_state(model, sd, strict=False)
    model = model.cuda().eval()
    from oracle.t5 import T5Oracle
    x = embeds(3, 19, geom.d_model)
    assert torch.equal(model.generate_from_embeds(x.cuda(), max_length=24).cpu(), T5Oracle(geom, sd).generate(x, 24))


@pytest.mark.parametrize("precision,B,S", [("bf16", 32, 190), ("fp32", 9, 61), ("bf16", 5, 864)])
def test_finished_row_early_out_does_not_change_ids(monkeypatch, precision, B, S):
    """Round 4: a (clip, head) workgroup of the decode attention kernels whose row has emitted EOS stops re-requesting K/V
    (csrc/decode.hip `row_fin`).  HF keeps computing such rows and forces their tokens to pad (hf generation/utils.py:2929), so
    nothing of them is observable: ids with the early-out (default) == ids without it (M2M_FINISHED_SKIP=0, a session of its own)
    == the oracle's (fp32), rows ending at different steps, one and two chains, K/V streamed non-temporally and not."""
    cfg = DEFAULT_CONFIG
    geom = T5Geometry(load_config(cfg).model.t5)
    sd = synth.t5_state_dict(geom, seed=0)
    synth.perturb_layer_norms(sd, 0)
    synth.force_eos_head(sd, geom, active=340, eos_scale=1.6)
    x = embeds(B, S, geom.d_model, seed=21)
    outs = {}
    for leg, env in (("on", None), ("off", "0")):
        if env is None:
            monkeypatch.delenv("M2M_FINISHED_SKIP", raising=False)
        else:
            monkeypatch.setenv("M2M_FINISHED_SKIP", env)
        m = T5Transformer(cfg, precision=precision)
        load_t5_state(m, sd, strict=False)
        m = m.cuda().eval()
        outs[leg] = m.generate_from_embeds(x.cuda(), max_length=1024).cpu()
        del m
    a = outs["on"]
    ends = [int((a[r] == geom.eos_token_id).float().argmax()) if (a[r] == geom.eos_token_id).any() else -1 for r in range(B)]
    print(f"early-out {precision} B={B} S={S}: output length {a.shape[1]}, EOS positions {sorted(ends)}")
    assert torch.equal(a, outs["off"])
    assert sum(1 for e in ends if e > 0) >= B // 2 and len(set(ends)) > 1        # the case does exercise finished rows
    if precision == "fp32":
        from oracle.t5 import T5Oracle
        assert torch.equal(a, T5Oracle(geom, sd).generate(x, 1024))


@pytest.mark.parametrize("cfg_name,B,S,Ld", [("tiny", 3, 19, 12), ("tiny", 2, 130, 131), ("full", 2, 190, 24), ("full", 5, 864, 40), ("full", 33, 40, 129), ("full", 32, 864, 8)])
def test_norm_gemm_fused_kernel_is_bit_identical_to_the_two_kernel_path(monkeypatch, cfg_name, B, S, Ld):
    """Round 5: RMSNorm fused into the following product — `norm_gemm_kernel` normalises a 128-row panel once into LDS
    and sweeps every column tile with it — must change NOTHING: same norm arithmetic and rounding point, same k order per output
    element as rmsnorm_kernel + gemm_kernel (M2M_NORM_GEMM=0).  The batched teacher-forced pass goes through every epilogue of the
    fused kernel (head-major q/k, transposed V, gated GELU in both weight interleaves, cross-q, cross-K/V with the final encoder norm
    as its prologue, fp32 lm_head): its logits and the greedy ids must be bit-identical with the switch on and off."""
    cfg = tiny_config() if cfg_name == "tiny" else DEFAULT_CONFIG
    g = T5Geometry(load_config(cfg).model.t5)
    x = embeds(B, S, g.d_model).cuda()
    dec = torch.from_numpy((synth.uniform01(11, "dec", B * Ld) * (g.vocab_size - 3)).astype(np.int64).reshape(B, Ld) + 3).cuda()
    dec[:, 0] = g.decoder_start_token_id
    out = {}
    for flag in ("force", "0"):          # "force": the fused kernel whatever the size (by default problems under 160 row blocks keep the two-kernel path)
        monkeypatch.setenv("M2M_NORM_GEMM", flag)
        monkeypatch.setenv("M2M_RESID_PANEL", flag)      # the row-panel residual products (attention output / feed-forward down projection) likewise
        model, _, _ = build(cfg, "bf16")                 # the switches are latched when a session is created (round 6): a model per leg
        out[flag] = (model.logits_from_embeds(x, dec).cpu(), model.generate_from_embeds(x, max_length=min(Ld, 24)).cpu())
        del model
    assert torch.isfinite(out["force"][0]).all()
    assert torch.equal(out["force"][0], out["0"][0]), f"fused norm+GEMM logits differ: max |d| {(out['force'][0] - out['0'][0]).abs().max():.3e}"
    assert torch.equal(out["force"][1], out["0"][1])


@pytest.mark.parametrize("cfg_name,B,S,Ld", [("tiny", 3, 19, 12), ("tiny", 2, 130, 131), ("full", 2, 190, 70), ("full", 3, 864, 40), ("full", 2, 61, 300), ("full", 9, 1000, 8), ("full", 2, 1100, 5), ("full", 2, 96, 33)])
def test_attention_wide_form_against_the_first_form(monkeypatch, cfg_name, B, S, Ld):
    """Round 5: the bf16 mode's flash attention takes 64 keys per softmax step from double-buffered tiles (`attn_wide_kernel`);
    the fp32 mode keeps the first kernel, 32 keys per step.  The two forms round the same probabilities at different running maxima
    and add the row sums in a different order, so they are NOT bit-identical; both are held to the oracle by the forced / golden
    tests, and here to each other through the batched teacher-forced pass (encoder self-attention with bias, S not a multiple of 64;
    decoder causal self-attention incl. Ld > 256: wide tiles below the diagonal, masked ones on it; cross-attention without bias):
    the logits may differ by bf16 re-rounding only (measured 4e-3 .. 7e-3 relative l2, the size of the bf16 mode's own noise floor against
    the emulating oracle) — bar 2 %; the fp32 mode must not change at all."""
    cfg = tiny_config() if cfg_name == "tiny" else DEFAULT_CONFIG
    g = T5Geometry(load_config(cfg).model.t5)
    x = embeds(B, S, g.d_model).cuda()
    dec = torch.from_numpy((synth.uniform01(12, "dec", B * Ld) * (g.vocab_size - 3)).astype(np.int64).reshape(B, Ld) + 3).cuda()
    dec[:, 0] = g.decoder_start_token_id
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("M2M_ATTN_WIDE", flag)
        model, _, _ = build(cfg, "bf16")                 # latched per session: a model per leg
        out[flag] = model.logits_from_embeds(x, dec).float().cpu()
        del model
    assert torch.isfinite(out["1"]).all()
    d = (out["1"] - out["0"]).double()
    rel = float(d.norm() / out["0"].double().norm())
    print(f"attention forms {cfg_name} B={B} S={S} Ld={Ld}: rel l2 {rel:.2e}, max |d| {float(d.abs().max()):.3e} on logits of max {float(out['0'].abs().max()):.1f}")
    if S >= 128:           # at least one 64-key step (shorter inputs take the masked 32-key steps in both forms)
        assert not torch.equal(out["1"], out["0"]), "the switch did not change the kernel"
    assert rel < 2e-2
    o32 = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("M2M_ATTN_WIDE", flag)
        m32, _, _ = build(cfg, "fp32")
        o32[flag] = m32.logits_from_embeds(x, dec).cpu()
        del m32
    assert torch.equal(o32["1"], o32["0"])


@pytest.mark.parametrize("precision,B,S", [("bf16", 32, 190), ("fp32", 9, 61), ("bf16", 5, 864), ("bf16", 128, 40), ("fp32", 33, 30)])
def test_live_row_repacking_does_not_change_ids(monkeypatch, precision, B, S):
    """Round 5: at a host poll of the greedy loop, once a quarter of the rows still being decoded have emitted EOS,
    the live rows are MOVED into the first slots (self K/V up to the current position, cross K/V, pending arg-max key, residual row;
    token rows stay, slot -> clip is a table) and smaller chains take over — merging two chains into one when the live rows fit.
    A row's arithmetic never depended on its slot, so ids with re-packing (default) == ids without (M2M_COMPACT=0) == the oracle's
    (fp32); the case must really re-pack (rows moved), and the session must decode a following full batch correctly
    (graphs of every view are cached, the slot table is reset by the next decode)."""
    cfg = DEFAULT_CONFIG
    geom = T5Geometry(load_config(cfg).model.t5)
    sd = synth.t5_state_dict(geom, seed=0)
    synth.perturb_layer_norms(sd, 0)
    synth.force_eos_head(sd, geom, active=340, eos_scale=1.6)
    x = embeds(B, S, geom.d_model, seed=21)
    m = T5Transformer(cfg, precision=precision)
    load_t5_state(m, sd, strict=False)
    m = m.cuda().eval()
    outs = {}
    for leg, env in (("on", "1"), ("off", "0"), ("on_again", "1")):
        monkeypatch.setenv("M2M_COMPACT", env)
        outs[leg] = m.generate_from_embeds(x.cuda(), max_length=1024).cpu()
        stats = m.repack_stats()
        if env == "1":
            print(f"re-packing {precision} B={B} S={S}: {stats[0]} re-packings, {stats[1]} rows moved, output length {outs[leg].shape[1]}")
            assert stats[0] >= 1 and stats[1] >= 1
        else:
            assert stats == (0, 0)
    assert torch.equal(outs["on"], outs["off"]) and torch.equal(outs["on_again"], outs["off"])
    a = outs["on"]
    ends = [int((a[r] == geom.eos_token_id).float().argmax()) if (a[r] == geom.eos_token_id).any() else -1 for r in range(B)]
    assert sum(1 for e in ends if e > 0) >= B // 2 and len(set(ends)) > 2
    if precision == "fp32":
        from oracle.t5 import T5Oracle
        assert torch.equal(a, T5Oracle(geom, sd).generate(x, 1024))
    # a batch without EOS on the same session afterwards: full-size chains again, nothing left over from the permuted slots
    sd2 = synth.t5_state_dict(geom, seed=0)
    synth.perturb_layer_norms(sd2, 0)
    m2 = T5Transformer(cfg, precision=precision)
    load_t5_state(m2, sd2, strict=False)
    m2 = m2.cuda().eval()
    want = m2.generate_from_embeds(x.cuda(), max_length=96).cpu()
    load_t5_state(m, sd2, strict=False)
    got = m.generate_from_embeds(x.cuda(), max_length=96).cpu()
    assert torch.equal(got, want)


def test_live_row_repacking_without_graphs_and_with_small_chains(monkeypatch):
    """The re-packing path under the other launch modes: direct launches instead of replayed graphs (M2M_NO_GRAPH=1) and four
    chains of 8 clips (M2M_GROUP_ROWS=8: re-planning then shrinks the NUMBER of chains, not only their width)."""
    cfg = DEFAULT_CONFIG
    geom = T5Geometry(load_config(cfg).model.t5)
    sd = synth.t5_state_dict(geom, seed=0)
    synth.perturb_layer_norms(sd, 0)
    synth.force_eos_head(sd, geom, active=340, eos_scale=1.6)
    x = embeds(32, 61, geom.d_model, seed=21)
    m = T5Transformer(cfg, precision="fp32")
    load_t5_state(m, sd, strict=False)
    m = m.cuda().eval()
    monkeypatch.setenv("M2M_COMPACT", "0")
    want = m.generate_from_embeds(x.cuda(), max_length=400).cpu()
    monkeypatch.setenv("M2M_COMPACT", "1")
    for env in ({"M2M_NO_GRAPH": "1"}, {"M2M_GROUP_ROWS": "8"}, {"M2M_GROUP_ROWS": "8", "M2M_NO_GRAPH": "1"}):
        for k in ("M2M_NO_GRAPH", "M2M_GROUP_ROWS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = m.generate_from_embeds(x.cuda(), max_length=400).cpu()
        stats = m.repack_stats()
        print(f"re-packing under {env}: {stats}")
        assert stats[0] >= 1 and torch.equal(got, want), env


@pytest.mark.parametrize("precision,B,S,L", [("bf16", 50, 61, 300), ("fp32", 13, 40, 200), ("bf16", 27, 190, 160), ("fp32", 49, 30, 120)])
def test_multi_clip_attention_and_wide_ff_tiles_are_bit_identical(monkeypatch, precision, B, S, L):
    """Round 6: large chains run the decode attention with C = 2 / 4 clips of one head per workgroup
    (dec_attn_mc_kernel: one row round trip and ONE fetch of the head's weights for C clips, the clips' K/V streams back to back)
    and the feed-forward with 2 / 4 hidden slices per workgroup summed as integers before ONE atomic (dec_ff_multi_kernel; also
    the 16-row tiles of the first form).  Per row the arithmetic is the first kernels', operation for operation: ids,
    output length and the teacher-forced step logits must be torch.equal whatever form a chain takes — with rows that end raggedly
    (early-out + re-packing on), chain sizes that are no multiple of C (clamped tail clips), one and two chains."""
    cfg = DEFAULT_CONFIG
    geom = T5Geometry(load_config(cfg).model.t5)
    sd = synth.t5_state_dict(geom, seed=0)
    synth.perturb_layer_norms(sd, 0)
    synth.force_eos_head(sd, geom, active=340, eos_scale=1.6)
    x = embeds(B, S, geom.d_model, seed=33)
    ids, logits = {}, {}
    base = ("1", "8", "1")
    for leg in (base, ("2", "16", "2"), ("4", "8", "4"), ("4", "16", "1"), ("2", "8", "4"), ("0", "0", "0")):
        clips, rows, slices = leg
        monkeypatch.setenv("M2M_DA_CLIPS", clips)           # latched when the session is created: a new model per leg
        monkeypatch.setenv("M2M_DEC_FF_ROWS", rows)
        monkeypatch.setenv("M2M_DEC_FF_SLICES", slices)
        m = T5Transformer(cfg, precision=precision)
        load_t5_state(m, sd, strict=False)
        m = m.cuda().eval()
        monkeypatch.delenv("M2M_FORWARD", raising=False)
        ids[leg] = m.generate_from_embeds(x.cuda(), max_length=L).cpu()
        monkeypatch.setenv("M2M_FORWARD", "step")           # the same step kernels, teacher-forced along the ids just decoded
        Ld = min(24, ids[leg].shape[1])
        logits[leg] = m.logits_from_embeds(x.cuda(), ids[base][:, :Ld].cuda()).cpu()
        del m
    for k in ids:
        assert torch.equal(ids[k], ids[base]), f"ids differ with M2M_DA_CLIPS={k[0]} M2M_DEC_FF_ROWS={k[1]} M2M_DEC_FF_SLICES={k[2]}"
        assert torch.equal(logits[k], logits[base]), f"step logits differ with M2M_DA_CLIPS={k[0]} M2M_DEC_FF_ROWS={k[1]} M2M_DEC_FF_SLICES={k[2]}"
    a = ids[base]
    ends = [int((a[r] == geom.eos_token_id).float().argmax()) if (a[r] == geom.eos_token_id).any() else -1 for r in range(B)]
    print(f"multi-clip {precision} B={B} S={S}: {sum(1 for e in ends if e > 0)} of {B} rows end, length {a.shape[1]}")
    if precision == "fp32":
        from oracle.t5 import T5Oracle
        assert torch.equal(a, T5Oracle(geom, sd).generate(x, L))


def test_encoder_switches_are_latched_when_the_session_is_created(monkeypatch):
    """M2M_NORM_GEMM / M2M_RESID_PANEL / M2M_ATTN_WIDE / M2M_NORM_GEMM_MIN_BLOCKS choose between kernel forms that are not
    all bit-identical; they are read ONCE per session, so an existing session keeps its kernels when the environment changes under it
    (and no launch calls getenv).  The attention forms differ in the last bits: an existing model must not notice the flip."""
    cfg = DEFAULT_CONFIG
    g = T5Geometry(load_config(cfg).model.t5)
    B, S, Ld = 2, 190, 40
    x = embeds(B, S, g.d_model).cuda()
    dec = torch.from_numpy((synth.uniform01(12, "dec", B * Ld) * (g.vocab_size - 3)).astype(np.int64).reshape(B, Ld) + 3).cuda()
    dec[:, 0] = g.decoder_start_token_id
    monkeypatch.setenv("M2M_ATTN_WIDE", "1")
    model, _, _ = build(cfg, "bf16")
    a = model.logits_from_embeds(x, dec).cpu()
    monkeypatch.setenv("M2M_ATTN_WIDE", "0")
    b = model.logits_from_embeds(x, dec).cpu()          # same session: still the wide form
    assert torch.equal(a, b)
    other, _, _ = build(cfg, "bf16")                    # a new session reads the environment again
    c = other.logits_from_embeds(x, dec).cpu()
    assert not torch.equal(a, c)


@pytest.mark.parametrize("d_model,d_ff,heads", [(512, 1024, 8), (256, 512, 4), (128, 256, 2)])
def test_other_model_widths_through_every_decode_form(monkeypatch, d_model, d_ff, heads):
    """m2m_model_create accepts d_model 128 / 256 / 384 / 512: the widths the reference never uses go through the same kernel forms
    (the multi-clip attention normalises a row with a wave PAIR — one wave at d_model <= 256 —, keeps 512 fixed-point columns per
    row in LDS, and the multi-slice feed-forward needs d_ff / 32 divisible by its slice count).  fp32 ids == oracle for the first
    forms and for C = 2 / 4 clips per workgroup with 2 / 4 slices, on a ragged-EOS batch that is no multiple of C."""
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    cfg["model"]["t5"].update(d_model=d_model, d_ff=d_ff, num_layers=2, num_decoder_layers=2, num_heads=heads)
    geom = T5Geometry(load_config(cfg).model.t5)
    sd = synth.t5_state_dict(geom, seed=3)
    synth.perturb_layer_norms(sd, 3)
    synth.force_eos_head(sd, geom)
    B, S, L = 11, 70, 48
    x = embeds(B, S, geom.d_model, seed=41)
    from oracle.t5 import T5Oracle
    want = T5Oracle(geom, sd).generate(x, L)
    for clips, slices in (("1", "1"), ("2", "2"), ("4", "4"), ("0", "0")):
        monkeypatch.setenv("M2M_DA_CLIPS", clips)
        monkeypatch.setenv("M2M_DEC_FF_SLICES", slices)
        m = T5Transformer(cfg, precision="fp32")
        load_t5_state(m, sd, strict=False)
        m = m.cuda().eval()
        got = m.generate_from_embeds(x.cuda(), max_length=L).cpu()
        assert torch.equal(got, want), f"d_model={d_model}: ids differ from the oracle with M2M_DA_CLIPS={clips} M2M_DEC_FF_SLICES={slices}"
        del m
