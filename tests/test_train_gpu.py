"""GPU parity of the training step: native forward + backward + Adafactor vs the oracle
(torch autograd over oracle/train.py, itself pinned to HuggingFace T5 + transformers' Adafactor by the
`train` golden) — ref: music2midi/model.py:27-43, transformer.py:28-39."""
import copy
import os

import numpy as np
import pytest
import torch

from music2midi_amd import synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import DEFAULT_CONFIG, T5Geometry, load_config
from music2midi_amd.input import ModelInputs
from music2midi_amd.transformer import T5Transformer

from test_t5_gpu import tiny_config

pytestmark = pytest.mark.gpu


def _setup(cfg, precision, B, F, Ld, seed=0, max_sizes=None):
    from music2midi_amd.training import NativeTrainer
    from oracle.train import T5TrainOracle, leaf_params
    geom = T5Geometry(load_config(cfg).model.t5)
    sd = synth.t5_state_dict(geom, seed=seed)
    synth.perturb_layer_norms(sd, seed)
    model = T5Transformer(cfg, precision="fp32")
    load_t5_state(model, sd, strict=False)
    model = model.cuda()
    tr = NativeTrainer(model, *(max_sizes or (B, F + 2, Ld)), precision=precision)
    feats = torch.from_numpy(synth.normal(5, "feats", (B, F, geom.d_model), 2.0))
    cond = torch.from_numpy(synth.cond_index_batch(2, B))
    labels = torch.from_numpy((synth.uniform01(4, "labels", B * Ld) * 330).astype(np.int64).reshape(B, Ld)) + 3
    if Ld > 6:
        labels[1 % B, Ld - 4:] = -100
        labels[(2 % B), Ld - 1:] = -100
    x = torch.zeros((B, F + 2, geom.d_model))
    x[:, 2:] = feats
    params = leaf_params(sd)
    return model, tr, T5TrainOracle(geom, params), params, geom, x, feats, cond, labels


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-20))


# (F = 400: 13 key tiles per stripe, the fused attention kernels near their largest shape; F = 530: past it, the unfused path)
@pytest.mark.parametrize("cfg_name,B,F,Ld", [("tiny", 3, 21, 14), ("tiny", 2, 70, 33), ("tiny", 1, 9, 1), ("full", 2, 40, 20), ("tiny", 2, 400, 37),
                                             ("tiny", 1, 530, 5)])
def test_fp32_loss_logits_and_every_gradient_match_autograd(cfg_name, B, F, Ld):
    cfg = tiny_config() if cfg_name == "tiny" else copy.deepcopy(DEFAULT_CONFIG)
    model, tr, orc, params, geom, x, feats, cond, labels = _setup(cfg, "fp32", B, F, Ld)
    loss, logits = tr.forward_backward(x.cuda(), cond.cuda(), labels.cuda(), want_logits=True)
    loss_o, logits_o, grads_o = orc.loss_and_grads(feats, cond, labels)
    assert abs(loss.item() - loss_o.item()) < 1e-4 * max(1.0, abs(loss_o.item()))
    assert (logits.cpu() - logits_o).abs().max() < 2e-3
    worst = {}
    for name, (off, shape) in tr.layout.items():
        g_dev = tr.grads[off:off + int(np.prod(shape))].view(shape).cpu()
        worst[name] = _rel(g_dev, grads_o[name])
    bad = {k: v for k, v in worst.items() if v > 1e-4}
    print(f"{cfg_name} B={B} F={F} Ld={Ld}: loss {loss.item():.6f} (oracle {loss_o.item():.6f}); worst gradient rel err "
          f"{max(worst.values()):.2e} over {len(worst)} tensors")
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:5]
    # .grad of the module's parameters ARE the flat buffer
    p = dict(model.named_parameters())["transformer.decoder.block.1.layer.1.EncDecAttention.k.weight"]
    assert p.grad is not None and _rel(p.grad.cpu(), grads_o["transformer.decoder.block.1.layer.1.EncDecAttention.k.weight"]) < 1e-4
    # deterministic: a second call gives the same bits
    g1 = tr.grads.clone()
    loss2, _ = tr.forward_backward(x.cuda(), cond.cuda(), labels.cuda())
    assert torch.equal(g1, tr.grads)


def test_three_steps_reproduce_the_huggingface_golden(golden_dir):
    """Losses of three consecutive steps and the parameters after them vs the fixture produced by HF T5 +
    transformers.optimization.Adafactor(warmup_init=True) (tests/golden/make_golden.py train)."""
    z = np.load(golden_dir / "train.npz")
    B, F, Ld = [int(v) for v in z["meta"]]
    model, tr, _, _, geom, x, feats, cond, labels = _setup(tiny_config(), "fp32", B, F, Ld)
    assert np.array_equal(labels.numpy(), z["labels"].astype(np.int64))
    keys = [str(k) for k in z["keys"]]
    losses = []
    for step in range(3):
        loss, logits = tr.forward_backward(x.cuda(), cond.cuda(), labels.cuda(), want_logits=(step == 0))
        losses.append(loss.item())
        if step == 0:
            assert np.abs(logits.cpu().numpy()[:, ::4] - z["logits_sample"]).max() < 2e-3
            for i, k in enumerate(keys):
                off, shape = tr.layout[k]
                g = tr.grads[off:off + int(np.prod(shape))].cpu().double()
                assert abs(g.norm().item() - z["grad_l2"][i]) <= 2e-4 * z["grad_l2"][i] + 1e-9, k
                assert np.abs(np.resize(g.numpy()[:8], 8) - z["grad_head"][i]).max() <= 1e-4 * np.abs(z["grad_head"][i]).max() + 1e-7, k
        tr.optimizer_step()
    assert np.abs(np.asarray(losses) - z["losses"]).max() < 2e-4, (losses, z["losses"])
    assert tr.step_count == 3
    for i, k in enumerate(keys):
        off, shape = tr.layout[k]
        p = tr.params[off:off + int(np.prod(shape))].cpu()
        assert abs(p.double().abs().sum().item() - z["param_abs_sum_after3"][i]) <= 1e-5 * z["param_abs_sum_after3"][i], k
        assert np.abs(np.resize(p.numpy()[:8], 8) - z["param_head_after3"][i]).max() < 2e-6, k


def test_adafactor_kernels_match_the_oracle_over_many_steps():
    """The optimizer alone on synthetic gradients (all tensor shapes of the model: matrices, vectors, the 32x8 bias
    tables), 12 steps — relative-step warm-up, factored second moments, update clipping."""
    from oracle.train import AdafactorOracle
    model, tr, _, params, geom, *_ = _setup(tiny_config(), "fp32", 2, 9, 4)
    p_ref = {k: v.detach().clone() for k, v in params.items()}
    opt = AdafactorOracle(p_ref)
    for step in range(12):
        grads = {}
        for k, (off, shape) in tr.layout.items():
            scale = 10.0 ** (step % 4 - 2)                              # 1e-2 .. 10: exercises the clipping both ways
            g = torch.from_numpy(synth.normal(step, k, shape, scale))
            grads[k] = g
            tr.grads[off:off + g.numel()].copy_(g.reshape(-1))
        tr.optimizer_step()
        opt.step(grads)
    worst = max(_rel(tr.params[off:off + int(np.prod(shape))].view(shape).cpu(), p_ref[k]) for k, (off, shape) in tr.layout.items())
    print(f"adafactor 12 steps: worst parameter rel err {worst:.2e}")
    assert worst < 2e-5
    # state export / import round trip continues identically
    state = tr.optimizer_state()
    before = tr.params.clone()
    tr.optimizer_step()
    after = tr.params.clone()
    tr.params.copy_(before)
    tr.load_optimizer_state(state)
    tr.optimizer_step()
    assert torch.equal(tr.params, after) and tr.step_count == 13


def test_bf16_gradients_track_the_fp32_oracle_at_the_reference_geometry():
    """Throughput mode (bf16 GEMM inputs, fp32 accumulate / residual / norms / softmax / loss), full model, a batch
    shaped like ref config.yaml (3 s segments -> S = 190): per-tensor direction and size of the gradient."""
    B, F, Ld = 4, 188, 48
    model, tr, orc, params, geom, x, feats, cond, labels = _setup(copy.deepcopy(DEFAULT_CONFIG), "bf16", B, F, Ld)
    loss, _ = tr.forward_backward(x.cuda(), cond.cuda(), labels.cuda())
    loss_o, _, grads_o = orc.loss_and_grads(feats, cond, labels)
    assert abs(loss.item() - loss_o.item()) < 2e-2 * abs(loss_o.item())
    cos_min, worst = 1.0, 0.0
    for name, (off, shape) in tr.layout.items():
        g = tr.grads[off:off + int(np.prod(shape))].cpu().double()
        r = grads_o[name].reshape(-1).double()
        if r.norm() < 1e-12:
            continue
        cos = float(torch.dot(g, r) / (g.norm() * r.norm() + 1e-30))
        cos_min = min(cos_min, cos)
        worst = max(worst, float((g - r).norm() / r.norm()))
    print(f"bf16 full model: loss {loss.item():.4f} vs fp32 oracle {loss_o.item():.4f}; min cosine {cos_min:.5f}, worst rel l2 {worst:.3e}")
    assert cos_min > 0.995 and worst < 0.1


def test_music2midi_training_surface_learns_and_serves_the_new_weights():
    """Music2MIDI.configure_optimizers / training_step / fit_batches (ref model.py:27-43, train.py:40-41) on waveforms +
    notes: the loss of a fixed batch goes down, and generate() afterwards runs on the UPDATED weights."""
    from music2midi_amd.model import Music2MIDI
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    cfg["dataloader"]["batch_size"] = 3
    m = Music2MIDI(cfg).cuda()
    m.train_precision = "fp32"
    notes = (np.array([[0.10, 0.40, 60, 80], [0.50, 1.00, 64, 80], [1.20, 1.90, 67, 80]]),
             np.array([[0.05, 0.30, 50, 80], [0.70, 1.10, 55, 80]]),
             np.array([[0.00, 2.90, 40, 80], [0.30, 0.80, 76, 80], [1.00, 1.40, 77, 80], [1.50, 1.70, 79, 80]]))
    wav = torch.from_numpy(synth.waveform_batch(40, 3, 48000, "music")).cuda()
    idx = torch.from_numpy(synth.cond_index_batch(40, 3)).cuda()
    batch = ModelInputs(input_waveform=wav, notes_batch=notes, cond_index=idx)
    before = m.model.generate(batch, max_length=12).clone()
    w0 = m.model.transformer.lm_head.weight.detach().clone()
    (opt,), (sched,) = m.configure_optimizers()
    losses = m.fit_batches([batch] * 200, optimizer=opt)
    assert losses[-1] < losses[0] - 0.05, (losses[0], losses[-1])
    assert m.global_step == 200 and m._trainer.step_count == 200 and 0 < sched.get_last_lr()[0] <= 2e-4
    assert not torch.equal(w0, m.model.transformer.lm_head.weight.detach())       # the module's parameters ARE the trained buffer
    assert m._trainer.dropout == pytest.approx(0.1)            # train() mode: T5Config.dropout_rate, as the reference trains
    m.eval()                                                   # eval(): dropout off -> the step's loss is the inference path's
    loss_eval = m.model(batch).loss                                                # inference-path forward on the new weights
    assert abs(loss_eval.item() - m.training_step(batch, 0).item()) < 1e-3 * max(1.0, loss_eval.item())
    assert m._trainer.dropout == 0.0
    after = m.model.generate(batch, max_length=12)
    assert after.shape[0] == 3 and (after[:, 0] == 1).all()
    sd = m.state_dict()
    assert torch.equal(sd["model.transformer.lm_head.weight"], m.model.transformer.lm_head.weight.detach())


@pytest.mark.parametrize("precision", ["fp32"])
def test_dropout_step_matches_autograd_with_the_same_masks(precision):
    """Dropout 0.1 at every place hf: modeling_t5.py has it.  The device's masks are a counter-based hash, so the oracle
    regenerates exactly the same masks (oracle/train.py DropoutMasks) and autograd must give the same loss and gradients."""
    from oracle.train import DropoutMasks
    B, F, Ld = 3, 21, 14
    model, tr, orc, params, geom, x, feats, cond, labels = _setup(tiny_config(), precision, B, F, Ld)
    tr.set_dropout(0.1, seed=1234)
    loss, logits = tr.forward_backward(x.cuda(), cond.cuda(), labels.cuda(), want_logits=True)
    loss_o, logits_o, grads_o = orc.loss_and_grads(feats, cond, labels, DropoutMasks(0.1, 1234, 0))
    loss_plain, _, _ = orc.loss_and_grads(feats, cond, labels)
    assert abs(loss_plain.item() - loss_o.item()) > 1e-3                      # the masks do something
    assert abs(loss.item() - loss_o.item()) < 1e-4 * abs(loss_o.item()), (loss.item(), loss_o.item(), loss_plain.item())
    assert (logits.cpu() - logits_o).abs().max() < 3e-3
    worst = max(_rel(tr.grads[off:off + int(np.prod(shape))].view(shape).cpu(), grads_o[name]) for name, (off, shape) in tr.layout.items())
    print(f"dropout 0.1: loss {loss.item():.6f} (oracle with the same masks {loss_o.item():.6f}, without dropout {loss_plain.item():.6f}); "
          f"worst gradient rel err {worst:.2e}")
    assert worst < 1e-4
    # the second call draws different masks (call index 1), reproducibly
    loss2, _ = tr.forward_backward(x.cuda(), cond.cuda(), labels.cuda())
    l2 = loss2.item()
    loss2_o, _, _ = orc.loss_and_grads(feats, cond, labels, DropoutMasks(0.1, 1234, 1))
    assert abs(l2 - loss2_o.item()) < 1e-4 * abs(loss2_o.item()) and abs(l2 - loss_o.item()) > 1e-4
    tr.set_dropout(0.1, seed=1234)
    loss3, _ = tr.forward_backward(x.cuda(), cond.cuda(), labels.cuda())
    assert loss3.item() == pytest.approx(loss_o.item(), rel=1e-4)
    # keep rate of a large mask is 1 - p
    m = DropoutMasks(0.1, 7, 0).mask(5, 200000)
    assert abs(float((m > 0).float().mean()) - 0.9) < 3e-3


def test_dropout_bf16_full_model_statistics():
    """bf16, full model, dropout on: the step runs, is reproducible for a fixed seed, differs between seeds, and its loss
    is within the spread dropout causes (not an exact check: bf16 rounding points differ from the fp32 oracle's)."""
    B, F, Ld = 4, 188, 48
    model, tr, orc, params, geom, x, feats, cond, labels = _setup(copy.deepcopy(DEFAULT_CONFIG), "bf16", B, F, Ld)
    from oracle.train import DropoutMasks
    tr.set_dropout(0.1, seed=5)
    a, _ = tr.forward_backward(x.cuda(), cond.cuda(), labels.cuda()); a = a.item(); ga = tr.grads.clone()
    tr.set_dropout(0.1, seed=5)
    b, _ = tr.forward_backward(x.cuda(), cond.cuda(), labels.cuda()); b = b.item()
    assert a == b and torch.equal(ga, tr.grads)
    tr.set_dropout(0.1, seed=6)
    c, _ = tr.forward_backward(x.cuda(), cond.cuda(), labels.cuda()); c = c.item()
    assert c != a
    ref, _, _ = orc.loss_and_grads(feats, cond, labels, DropoutMasks(0.1, 5, 0))
    assert abs(a - ref.item()) < 3e-2 * abs(ref.item())


def fp8_agreement(tr, ref):
    cs, ws = [], []
    for name, (off, shape) in tr.layout.items():
        g = tr.grads[off:off + int(np.prod(shape))].cpu().double()
        r = ref[name].reshape(-1).double()
        if r.norm() < 1e-9:
            continue
        cs.append(float(torch.dot(g, r) / (g.norm() * r.norm() + 1e-30)))
        ws.append(float((g - r).norm() / r.norm()))
    return min(cs), float(np.median(cs)), max(ws)


def fp8_self_consistency(orc, feats, cond, labels, grads_emul):
    """How far the emulating oracle agrees WITH ITSELF when its input moves by 1e-4 (relative, seeded noise; a fortieth of a bf16
    ulp): an MXFP8 element has 3 mantissa bits, so a difference far below bf16 rounding flips a few quantiser decisions per layer,
    each flip moves its element by 6 %, and the flips compound with depth — measured on the host (full width, 4 clips, 1e-6 noise):
    per-tensor gradient cosine min / median 0.995 / 0.997 with 1 + 1 layers, 0.979 / 0.989 with 2 + 2, 0.953 / 0.964 with 6 + 6
    (1e-4 and 1e-3 noise give the same 0.945-0.951 / 0.956-0.960: the floor is reached by ANY difference), while the fp32 network
    under 1e-3 noise stays at 0.999995 and the bf16-emulating one at 0.9996 (rel l2 2.7e-2 — exactly what the device's bf16 mode
    shows against it).  Two evaluations of the fp8 step that are not BIT-identical in every quantiser input (device vs emulation:
    different fp32 summation orders, the hardware exp2) therefore cannot agree better than this floor; that, not a device /
    emulation mismatch, is where round 2's 0.93-0.95 came from."""
    noise = torch.from_numpy(synth.normal(77, "fp8_floor", tuple(feats.shape), 1.0))
    _, _, g2 = orc.loss_and_grads(feats * (1.0 + 1e-4 * noise), cond, labels)
    cs = []
    for k, r in grads_emul.items():
        if r.norm() < 1e-9:
            continue
        a, b = g2[k].reshape(-1).double(), r.reshape(-1).double()
        cs.append(float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)))
    return min(cs), float(np.median(cs))


def one_layer_config():
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    cfg["model"]["t5"].update(num_layers=1, num_decoder_layers=1)
    return cfg


@pytest.mark.parametrize("parts", [None, "fwd,dx,dw"])
@pytest.mark.parametrize("cfg_name,B,F,Ld", [("tiny", 3, 21, 14), ("full_1layer", 4, 188, 48), ("full", 4, 188, 48)])
def test_fp8_mode_matches_the_mx_emulating_oracle(cfg_name, B, F, Ld, parts, monkeypatch):
    """fp8 mode (BASELINE configs[4]'s dtype): the projection products — forward and dX by default, dW with M2M_FP8_PARTS=fwd,dx,dw
    (second parametrisation) — on block-scaled OCP FP8 (MXFP8 e4m3, 32 elements per power-of-two scale; csrc/mx8.hip), everything
    else as the bf16 mode.  The product itself is pinned by tests/test_mx8_gpu.py (element quantisation bit-identical).  The STEP is
    compared with autograd over an oracle whose three products per projection (forward, dX, dW) each quantise their own operands
    along their own reduction dimension, from values rounded to bfloat16 where the device stores bfloat16 (oracle/train.py _MxLinear,
    _Round): the arithmetic the device runs.  Two bars:
      * full width, ONE encoder + ONE decoder layer — where quantiser flips cannot compound — min cosine > 0.98 (the per-layer
        arithmetic is right: this is the localisation round 2's review asked for);
      * any depth: the device agrees with the emulation at least as well as the emulation agrees with itself under a 1e-4 input
        perturbation (fp8_self_consistency; - 0.015): at 6 + 6 layers that floor is ~0.95 / 0.96 and no implementation can beat it."""
    cfg = tiny_config() if cfg_name == "tiny" else one_layer_config() if cfg_name == "full_1layer" else copy.deepcopy(DEFAULT_CONFIG)
    if parts:
        monkeypatch.setenv("M2M_FP8_PARTS", parts)
    model, tr, orc, params, geom, x, feats, cond, labels = _setup(cfg, "fp8", B, F, Ld)
    loss, _ = tr.forward_backward(x.cuda(), cond.cuda(), labels.cuda())
    g1 = tr.grads.clone()
    loss2, _ = tr.forward_backward(x.cuda(), cond.cuda(), labels.cuda())
    assert torch.equal(g1, tr.grads) and loss.item() == loss2.item()                  # deterministic
    loss_plain, _, grads_plain = orc.loss_and_grads(feats, cond, labels)
    orc.mx8, orc.mx8_dw, orc.bf16 = True, bool(parts), True
    loss_o, _, grads_o = orc.loss_and_grads(feats, cond, labels)
    fmin, fmed = fp8_self_consistency(orc, feats, cond, labels, grads_o)
    cmin, cmed, worst = fp8_agreement(tr, grads_o)
    pmin, pmed, _ = fp8_agreement(tr, grads_plain)
    print(f"fp8 {cfg_name}: loss {loss.item():.4f} (MX-emulating oracle {loss_o.item():.4f}, fp32 oracle {loss_plain.item():.4f}); gradient cosine vs "
          f"the emulating oracle min {cmin:.4f} / median {cmed:.4f} (worst rel l2 {worst:.3f}); the emulation vs itself under 1e-4 input noise "
          f"min {fmin:.4f} / median {fmed:.4f}; vs the unquantised oracle min {pmin:.4f} / median {pmed:.4f}")
    assert abs(loss.item() - loss_o.item()) < 1e-2 * abs(loss_o.item())
    assert abs(loss.item() - loss_plain.item()) < 3e-2 * abs(loss_plain.item())
    assert cmed > fmed - 0.015 and cmin > fmin - 0.03      # (the minimum over the tensors is itself a noisy draw: wider)
    # ABSOLUTE floors beside the relative bar: a regression that makes device AND emulation noisier, or an emulation
    # that follows a device bug, must not pass on the relative bar alone.  Against the MX emulation (measured 0.981 / 0.987 tiny,
    # 0.995 / 0.997 one layer, 0.946-0.950 / 0.958-0.960 full) and — the independent check, an oracle with no device-mirroring
    # flags at all — against plain fp32 autograd (measured min / median 0.953 / 0.968, 0.972 / 0.982, 0.915 / 0.931).
    abs_floor = {"tiny": (0.97, 0.98, 0.94, 0.955), "full_1layer": (0.98, 0.99, 0.96, 0.97), "full": (0.93, 0.95, 0.90, 0.92)}[cfg_name]
    assert cmin > abs_floor[0] and cmed > abs_floor[1], (cmin, cmed, abs_floor)
    assert pmin > abs_floor[2] and pmed > abs_floor[3], (pmin, pmed, abs_floor)
    if cfg_name == "tiny":                            # and it trains (Adafactor's warm-up steps are ~1e-6: the first few do not
        loss0 = loss.item()                           # move an FP8-quantised weight at all, so give it a while); NB `loss` is
        for _ in range(150):                          # the trainer's own device scalar, overwritten by every call
            tr.optimizer_step()
            l, _ = tr.forward_backward(x.cuda(), cond.cuda(), labels.cuda())
        assert l.item() < loss0 - 0.05, (loss0, l.item())


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_graph_replay_equals_direct_issue(precision, monkeypatch):
    """The captured HIP graph of the step (second call with the same buffers and shapes onwards) must be the direct issue
    bit for bit — fresh inputs every call, dropout on (the mask sequence advances on the device) — and the one grouped
    weight-gradient launch must agree with the classic per-product split-K path to fp32 rounding."""
    from music2midi_amd.training import NativeTrainer
    B, F, Ld = 3, 21, 14
    model, tr_graph, orc, params, geom, x, feats, cond, labels = _setup(tiny_config(), precision, B, F, Ld)
    monkeypatch.setenv("M2M_TRAIN_GRAPH", "0")
    tr_direct = NativeTrainer(model, B, F + 2, Ld, precision=precision)        # adopts the same parameter values
    monkeypatch.setenv("M2M_TRAIN_DW_GROUP", "0")
    monkeypatch.setenv("M2M_TRAIN_SIDE", "0")
    tr_serial = NativeTrainer(model, B, F + 2, Ld, precision=precision)
    for tr in (tr_graph, tr_direct, tr_serial):
        tr.set_dropout(0.1, seed=77)
    worst = 0.0
    for call in range(5):                                                        # calls 1.. of tr_graph replay the graph
        xi = (x + 0.01 * call * torch.from_numpy(synth.normal(40 + call, "dx", tuple(x.shape), 1.0))).cuda()
        lab = labels.clone()
        lab[0, call % Ld] = 5 + call
        outs = []
        for tr in (tr_graph, tr_direct, tr_serial):
            loss, _ = tr.forward_backward(xi, cond.cuda(), lab.cuda())
            outs.append((loss.item(), tr.grads.clone()))
        assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1]), f"call {call}: graph replay differs from direct issue"
        assert abs(outs[0][0] - outs[2][0]) <= 1e-6 * abs(outs[2][0])
        tol = 1e-5 if precision == "fp32" else 1e-4
        worst = max(worst, _rel(outs[0][1], outs[2][1]))
        assert worst < tol, (call, worst)
        if call:
            assert outs[0][0] != first_loss                                      # inputs and masks did change
        first_loss = outs[0][0]
    print(f"{precision}: graph == direct over 5 calls; grouped vs split-K weight gradients {worst:.2e}")
    for tr in (tr_graph, tr_direct, tr_serial):
        tr.close()


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp8"])
def test_split_backward_releases_final_decoder_gradients_early(precision):
    """Data-parallel overlap (m2m_trainer_set_sync_stream): the backward pass issued in two parts must give the gradients of the
    unsplit pass bit for bit — direct issue, capture and replay — and the two early ranges (shared embedding + lm_head, decoder
    blocks) must be FINAL when the sync stream is released: a copy taken on that stream right after the call (it runs beside the
    encoder-side backward) equals the finished buffer, and together with the late ranges the pieces tile the buffer."""
    from music2midi_amd import distributed as D
    from music2midi_amd.training import NativeTrainer
    fp8 = precision == "fp8"
    cfg = tiny_config()
    if fp8:
        cfg = copy.deepcopy(cfg)
        cfg["model"]["t5"].update(d_model=128, d_ff=256)
    B, F, Ld = 3, 37, 19
    model, tr_split, orc, params, geom, x, feats, cond, labels = _setup(cfg, precision, B, F, Ld)
    tr_whole = NativeTrainer(model, B, F + 2, Ld, precision=precision)
    sync = torch.cuda.Stream()
    tr_split.set_sync_stream(sync)
    early, late = D.split_ranges(tr_split.n_floats, tr_split.early_ranges)
    assert len(early) == 2 and sum(c for _, c in early + late) == tr_split.n_floats
    names_early = [n for n, (off, _) in tr_split.layout.items() if any(o <= off < o + c for o, c in early)]
    assert any("decoder.block.0" in n for n in names_early) and any("lm_head" in n for n in names_early)
    assert not any("encoder." in n or "conditioning" in n for n in names_early), names_early[:4]
    for tr in (tr_split, tr_whole):
        tr.set_dropout(0.1, seed=11)
    for call in range(4):                                                        # call 0 direct, call 1 captures, 2.. replay
        xi = (x + 0.01 * call * torch.from_numpy(synth.normal(60 + call, "dx", tuple(x.shape), 1.0))).cuda()
        tr_split.grads.fill_(float("nan"))                                       # a range nobody wrote would show
        loss_s, _ = tr_split.forward_backward(xi, cond.cuda(), labels.cuda())
        with torch.cuda.stream(sync):                                            # what an all-reduce on the sync stream would read
            snap = [tr_split.grads[o:o + c].clone() for o, c in early]
        loss_w, _ = tr_whole.forward_backward(xi, cond.cuda(), labels.cuda())
        torch.cuda.synchronize()
        assert loss_s.item() == loss_w.item(), call
        assert torch.equal(tr_split.grads, tr_whole.grads), f"call {call}: split pass differs from the whole pass"
        for (o, c), sn in zip(early, snap):
            assert torch.equal(sn, tr_split.grads[o:o + c]), f"call {call}: range ({o}, {c}) was not final at the release"
    tr_split.set_sync_stream(None)                                               # and off again: one graph, same numbers
    for call in range(3):
        loss_s, _ = tr_split.forward_backward(x.cuda(), cond.cuda(), labels.cuda())
        loss_w, _ = tr_whole.forward_backward(x.cuda(), cond.cuda(), labels.cuda())
        torch.cuda.synchronize()
        assert loss_s.item() == loss_w.item() and torch.equal(tr_split.grads, tr_whole.grads)
    tr_split.close(); tr_whole.close()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_a_graph_per_shape_survives_other_shapes_in_between(precision, monkeypatch):
    """Labels are padded to the longest sequence of the batch, so (B, S, L) changes from step to step and comes back.  The grouped
    weight-gradient launch reads a device-resident table of operand pointers and reduction lengths; every cached shape owns its
    own table slot, so a captured graph replays ITS table whatever ran in between (round 2 kept one table: A, A, B, A replayed
    A's graph over B's table).  Three shapes interleaved, A A B A B B A C A C B A — every call bit-identical to a trainer
    that never captures."""
    from music2midi_amd.training import NativeTrainer
    shapes = {"A": (3, 21, 14), "B": (3, 21, 9), "C": (2, 30, 14)}
    Bm, Fm, Lm = 3, 30, 14
    model, tr_graph, orc, params, geom, x, feats, cond, labels = _setup(tiny_config(), precision, Bm, Fm, Lm)
    monkeypatch.setenv("M2M_TRAIN_GRAPH", "0")
    tr_direct = NativeTrainer(model, Bm, Fm + 2, Lm, precision=precision)
    for tr in (tr_graph, tr_direct):
        tr.set_dropout(0.1, seed=3)
    for call, name in enumerate("AABABBACACBA"):
        b, f, l = shapes[name]
        xi = (x[:b, :f + 2] + 0.01 * call).contiguous().cuda()
        ci, li = cond[:b].cuda(), labels[:b, :l].contiguous().cuda()
        la, _ = tr_graph.forward_backward(xi, ci, li)
        ga = tr_graph.grads.clone()
        lb, _ = tr_direct.forward_backward(xi, ci, li)
        assert la.item() == lb.item() and torch.equal(ga, tr_direct.grads), f"call {call} (shape {name}): graph path differs from direct issue"
    tr_graph.close(); tr_direct.close()


def test_more_shapes_than_graph_slots_recycles_the_least_recently_used(monkeypatch):
    from music2midi_amd.training import NativeTrainer
    model, tr_graph, orc, params, geom, x, feats, cond, labels = _setup(tiny_config(), "fp32", 3, 21, 14)
    monkeypatch.setenv("M2M_TRAIN_GRAPH", "0")
    tr_direct = NativeTrainer(model, 3, 23, 14, precision="fp32")
    order = list(range(3, 14)) * 2 + [3, 3, 13, 13, 3]          # 11 label lengths (> 8 slots), twice round, then revisits
    for call, l in enumerate(order):
        li = labels[:, :l].contiguous().cuda()
        la, _ = tr_graph.forward_backward(x.cuda(), cond.cuda(), li)
        ga = tr_graph.grads.clone()
        lb, _ = tr_direct.forward_backward(x.cuda(), cond.cuda(), li)
        assert la.item() == lb.item() and torch.equal(ga, tr_direct.grads), f"call {call} (L = {l})"
    tr_graph.close(); tr_direct.close()


@pytest.mark.parametrize("env", [{"M2M_TRAIN_DW_GROUP": "0"}, {"M2M_TRAIN_DW_GROUP": "0", "M2M_TRAIN_SIDE": "0"}])
def test_sync_stream_without_a_split_pass_is_released_behind_the_whole_pass(env, monkeypatch):
    """A sync stream is set but the pass cannot be split (per-product weight gradients): the stream must be released at the END of
    the pass, so the early all-reduce a caller enqueues on it reads finished gradients (round 2 never released it: a race)."""
    from music2midi_amd import distributed as D
    from music2midi_amd.training import NativeTrainer
    B, F, Ld = 3, 37, 19
    model, tr_ref, orc, params, geom, x, feats, cond, labels = _setup(tiny_config(), "bf16", B, F, Ld)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    tr = NativeTrainer(model, B, F + 2, Ld, precision="bf16")
    sync = torch.cuda.Stream()
    tr.set_sync_stream(sync)
    early, _ = D.split_ranges(tr.n_floats, tr.early_ranges)
    for call in range(3):
        tr.grads.fill_(float("nan"))
        tr.forward_backward(x.cuda(), cond.cuda(), labels.cuda())
        with torch.cuda.stream(sync):
            snap = [tr.grads[o:o + c].clone() for o, c in early]
        tr_ref.forward_backward(x.cuda(), cond.cuda(), labels.cuda())
        torch.cuda.synchronize()
        assert not torch.isnan(tr.grads).any()
        for (o, c), sn in zip(early, snap):
            assert torch.equal(sn, tr.grads[o:o + c]), f"call {call}: the sync stream ran ahead of the pass"
        # per-product split-K weight gradients vs the grouped launch: same numbers to rounding
        assert _rel(tr.grads, tr_ref.grads) < 1e-3
    tr.close(); tr_ref.close()


def test_a_batch_without_any_scored_label_gives_nan_loss_and_zero_gradients():
    """torch / HF CrossEntropyLoss(ignore_index=-100) over zero scored rows is NaN with a zero gradient; so is the device's."""
    model, tr, orc, params, geom, x, feats, cond, labels = _setup(tiny_config(), "fp32", 2, 9, 6)
    lab = torch.full_like(labels, -100)
    loss, _ = tr.forward_backward(x.cuda(), cond.cuda(), lab.cuda())
    loss_o, _, grads_o = orc.loss_and_grads(feats, cond, lab)
    assert torch.isnan(loss_o) and all(float(g.abs().max()) == 0.0 for g in grads_o.values())
    assert torch.isnan(loss).all() and float(tr.grads.abs().max()) == 0.0
    tr.close()


def _notes_batches(n_batches, B=3):
    out = []
    for i in range(n_batches):
        notes = []
        for b in range(B):
            n = 2 + (5 * i + 3 * b) % 7
            u = synth.uniform01(100 + i, f"notes{b}", n * 3).reshape(n, 3)
            on = np.sort(u[:, 0] * 2.5)
            notes.append(np.stack([on, on + 0.05 + u[:, 1] * 0.4, np.floor(40 + u[:, 2] * 40), np.full(n, 80.0)], axis=1))
        wav = torch.from_numpy(synth.waveform_batch(200 + 3 * i, B, 48000, "music")).cuda()
        idx = torch.from_numpy(synth.cond_index_batch(7 + i, B)).cuda()
        out.append(ModelInputs(input_waveform=wav, notes_batch=tuple(notes), cond_index=idx))
    return out


def test_fit_batches_resumes_from_a_checkpoint_bit_for_bit(tmp_path):
    """ref train.py:41 ``trainer.fit(..., ckpt_path=)``: k steps, a Lightning-layout .ckpt (weights + Adafactor state in
    transformers' own state layout + step counter), a NEW process-worth of objects loading it and continuing m steps ==
    the uninterrupted k + m steps, bit for bit (dropout on: the mask sequence continues from the restored step)."""
    from music2midi_amd.checkpoint import read_checkpoint
    from music2midi_amd.model import Music2MIDI
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    cfg["dataloader"]["batch_size"] = 3
    cfg["trainer"]["log_every_n_steps"] = 1000                  # no greedy decode inside the steps of this test
    batches = _notes_batches(7)
    k = 4

    def fresh():
        torch.manual_seed(5)
        m = Music2MIDI(copy.deepcopy(cfg)).cuda()
        m.train_precision = "bf16"
        return m

    whole = fresh()
    losses_whole = whole.fit_batches(batches)
    first = fresh()
    ck = tmp_path / "step4.ckpt"
    losses_a = first.fit_batches(batches[:k], save_path=ck)
    raw = read_checkpoint(ck)
    assert raw["global_step"] == k and raw["hyper_parameters"]["config_path"] is not None
    opt = raw["optimizer_states"][0]
    n_params = len(list(first.parameters()))
    assert sorted(opt["state"]) == list(range(n_params)) and opt["param_groups"][0]["params"] == list(range(n_params))
    some = opt["state"][1]                                      # a matrix: factored second moments, HF's names
    assert set(some) >= {"step", "exp_avg_sq_row", "exp_avg_sq_col", "RMS"} and some["step"] == k
    second = Music2MIDI(copy.deepcopy(cfg)).cuda()              # different random init: everything must come from the file
    second.train_precision = "bf16"
    losses_b = second.fit_batches(batches[k:], ckpt_path=ck)
    assert second.global_step == len(batches) and second._trainer.step_count == len(batches)
    assert losses_a + losses_b == losses_whole, (losses_a + losses_b, losses_whole)
    for (n1, p1), (n2, p2) in zip(whole.named_parameters(), second.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2), n1
    assert torch.equal(whole._trainer.optimizer_state()["second_moments"], second._trainer.optimizer_state()["second_moments"])
    # load_from_checkpoint reads the same file for inference
    third = Music2MIDI.load_from_checkpoint(ck, config_path=copy.deepcopy(cfg)).cuda()
    assert torch.equal(third.model.transformer.lm_head.weight, first.model.transformer.lm_head.weight)


def test_training_step_pads_labels_to_a_bucket_without_changing_loss_or_gradients():
    """Music2MIDI.training_step rounds the label length up to LABEL_BUCKET with ignored positions (so that the trainer meets few
    shapes): same loss, same gradients as the batch's own length, and nothing in the step waits for the device."""
    from music2midi_amd.model import Music2MIDI
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    cfg["dataloader"]["batch_size"] = 3
    cfg["trainer"]["log_every_n_steps"] = 1000
    m = Music2MIDI(cfg).cuda().eval()                           # eval(): dropout off, so the two passes are comparable
    m.train_precision = "fp32"
    batch = _notes_batches(1)[0]
    loss = m.training_step(batch, 0)
    assert torch.is_tensor(loss) and loss.is_cuda and torch.is_tensor(m.logged["train/loss"])
    g_bucket = m._trainer.grads.clone()
    raw = m.model.tokenizer(batch.notes_batch)
    assert raw.shape[1] % m.LABEL_BUCKET != 0                   # the batch really is padded
    raw[raw == 0] = -100
    x = m.model.encoder_inputs(batch)
    loss_raw, _ = m._trainer.forward_backward(x, batch.cond_index, raw)
    assert abs(loss.item() - loss_raw.item()) < 1e-6 * abs(loss_raw.item())
    assert _rel(g_bucket, m._trainer.grads) < 1e-5
    logged = m.logged_metrics()
    assert logged["train/loss"] == pytest.approx(loss.item()) and logged["batch_size"] == 3


@pytest.mark.parametrize("cfg_name,B,F,Ld,p", [("tiny", 3, 21, 14, 0.0), ("tiny", 2, 70, 33, 0.1), ("full", 4, 188, 48, 0.1), ("full", 8, 259, 256, 0.1)])
def test_whole_head_attention_step_against_autograd_and_the_stripe_path(cfg_name, B, F, Ld, p, monkeypatch):
    """The bf16 step with the whole-head attention kernels (csrc/attn_train.hip: no stored probabilities; log-sum-exp, keep-bit
    words and recomputation) and the same step on round 2's stripe kernels (M2M_TRAIN_ATTN=stripes), same weights, inputs and
    dropout masks.  Both are held against fp32 autograd over the SAME masks (oracle/train.py regenerates them from the hash): a
    wrong mask, bias diagonal or row statistic in either path shows there as a cosine far below 0.99.  Against each other they can
    only agree to the bf16 floor — two bf16 evaluations of twelve layers that round at different points differ by ~3e-2 rel l2
    (fp8_self_consistency's docstring has the measurement) — so that comparison is a bound on the median, not a parity claim."""
    from music2midi_amd.training import NativeTrainer
    from oracle.train import DropoutMasks
    cfg = tiny_config() if cfg_name == "tiny" else copy.deepcopy(DEFAULT_CONFIG)
    monkeypatch.setenv("M2M_TRAIN_GRAPH", "0")
    torch.set_num_threads(min(16, os.cpu_count() or 16))
    model, tr_head, orc, params, geom, x, feats, cond, labels = _setup(cfg, "bf16", B, F, Ld)
    tr_stripe = NativeTrainer(model, B, F + 2, Ld, precision="bf16")
    for tr in (tr_head, tr_stripe):
        if p:
            tr.set_dropout(p, seed=31)
    monkeypatch.setenv("M2M_TRAIN_ATTN", "head")
    la, _ = tr_head.forward_backward(x.cuda(), cond.cuda(), labels.cuda())
    la = la.item()
    monkeypatch.setenv("M2M_TRAIN_ATTN", "stripes")
    lb, _ = tr_stripe.forward_backward(x.cuda(), cond.cuda(), labels.cuda())
    lb = lb.item()
    loss_o, _, grads_o = orc.loss_and_grads(feats, cond, labels, DropoutMasks(p, 31, 0) if p else None)

    def against(tr):
        cs, ws = [], []
        for name, (off, shape) in tr.layout.items():
            g, ref = tr.grads[off:off + int(np.prod(shape))].cpu().double(), grads_o[name].reshape(-1).double()
            if float(ref.norm()) < 1e-12:
                continue
            cs.append(float(torch.dot(g, ref) / (g.norm() * ref.norm() + 1e-30)))
            ws.append(float((g - ref).norm() / ref.norm()))
        return min(cs), max(ws)
    ch, wh = against(tr_head)
    cst, wst = against(tr_stripe)
    errs = []
    for name, (off, shape) in tr_head.layout.items():
        n = int(np.prod(shape))
        a, b = tr_head.grads[off:off + n].double(), tr_stripe.grads[off:off + n].double()
        if float(b.norm()) < 1e-12:
            continue
        errs.append((float((a - b).norm() / b.norm()), name))
    worst, worst_name = max(errs)
    med = float(np.median([e for e, _ in errs]))
    print(f"{cfg_name} B={B} F={F} Ld={Ld} dropout {p}: loss {la:.5f} (whole-head) / {lb:.5f} (stripes) / {loss_o.item():.5f} (fp32 autograd, same masks); "
          f"gradients vs autograd: whole-head cosine min {ch:.5f}, worst rel l2 {wh:.3f}; stripes {cst:.5f}, {wst:.3f}; "
          f"whole-head vs stripes rel l2 median {med:.2e}, worst {worst:.2e} ({worst_name})")
    tol = 3e-2 if cfg_name == "tiny" else 2e-2                   # (tiny: 32-wide tensors and loss ~ 30: a handful of roundings decide)
    assert abs(la - loss_o.item()) < tol * abs(loss_o.item()) and abs(lb - loss_o.item()) < tol * abs(loss_o.item())
    assert ch > (0.99 if cfg_name == "tiny" else 0.995) and wh < 0.12
    assert wh < 1.3 * wst + 1e-2                                 # no further from autograd than the stored-probability path is
    assert med < 4e-2 and worst < 8e-2
    tr_head.close(); tr_stripe.close()


def test_trainer_rejects_more_label_positions_than_the_embedding_gradient_lists():
    """embed_bwd_kernel keeps the pass's id list in LDS (4 B per label position + 16 KiB, 158 KiB opt-in); a trainer
    sized beyond that failed at its first launch with a generic HIP error.  Now m2m_trainer_create says so (64 x 640 labels), and
    the size just inside the limit is accepted."""
    from music2midi_amd import native
    from music2midi_amd.training import NativeTrainer
    cfg = tiny_config()
    model = T5Transformer(cfg, precision="fp32").cuda()
    with pytest.raises(native.NativeError, match="label positions per pass"):
        NativeTrainer(model, 64, 23, 640, precision="bf16")
    NativeTrainer(model, 64, 23, 560, precision="bf16").close()          # 35 840 positions <= 36 352
