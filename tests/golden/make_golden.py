#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ (run in the BUILD container only).

    python tests/golden/make_golden.py

The reference has no tests or fixtures of its own, so the vectors
are produced here by the real third-party code the reference calls, as far as it
is importable in this container:

* HuggingFace ``T5ForConditionalGeneration`` (transformers 5.15.0) forced to
  eager attention with an untied ``lm_head`` (= the pinned 4.34.0 semantics,
  SURVEY.md §0.3), built from the reference's own ``config.yaml`` ``model.t5``
  section  -> encoder states, teacher-forced logits, greedy ids.
* ``torch.stft`` for the STFT half of the frontend; the mel filterbank half is the
  restated torchaudio formula (torchaudio is absent: "filterbank parity unpinned").
* The reference's ``music2midi/tokenizer.py`` itself, imported from
  /root/reference under three shims (numba.njit -> identity, omegaconf.DictConfig
  -> dict, np.float_ -> np.float64) with bytecode writing disabled.

Only inputs/outputs are stored (weights and waveforms are regenerated from
music2midi_amd.synth seeds); no reference source text is copied.
"""
from __future__ import annotations

import importlib.util
import json
import sys
import types
from pathlib import Path

sys.dont_write_bytecode = True
HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

from music2midi_amd import synth  # noqa: E402
from music2midi_amd.config import ConfigNode, T5Geometry  # noqa: E402
from oracle.logmel import LogMelOracle, melscale_fbanks  # noqa: E402
from oracle.t5 import T5Oracle  # noqa: E402

REF = Path("/root/reference")


# ------------------------------------------------------------------ helpers
def ref_config() -> dict:
    return yaml.safe_load((REF / "config.yaml").read_text())


def tiny_t5(cfg: dict) -> dict:
    t5 = dict(cfg["model"]["t5"])
    t5.update(d_model=128, d_ff=256, num_layers=2, num_decoder_layers=2, num_heads=2)
    return t5


def build_hf(t5cfg: dict, sd: dict):
    from transformers import T5Config, T5ForConditionalGeneration
    cfg = T5Config(**t5cfg)
    cfg._attn_implementation = "eager"
    m = T5ForConditionalGeneration(cfg)
    m.config.tie_word_embeddings = False
    m.lm_head.weight = torch.nn.Parameter(torch.zeros_like(m.lm_head.weight))   # untie (4.34.0)
    hf_sd = {k[len("transformer."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("transformer.")}
    hf_sd["encoder.embed_tokens.weight"] = hf_sd["shared.weight"]
    hf_sd["decoder.embed_tokens.weight"] = hf_sd["shared.weight"]
    missing, unexpected = m.load_state_dict(hf_sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    assert m.lm_head.weight.data_ptr() != m.shared.weight.data_ptr()
    assert getattr(m.config, "scale_decoder_outputs", False) is False
    return m.eval()


def import_reference_tokenizer():
    numba = types.ModuleType("numba")
    numba.njit = lambda f=None, **k: (f if f is not None else (lambda g: g))
    omegaconf = types.ModuleType("omegaconf")
    omegaconf.DictConfig = dict
    sys.modules.setdefault("numba", numba)
    sys.modules.setdefault("omegaconf", omegaconf)
    if not hasattr(np, "float_"):
        np.float_ = np.float64
    spec = importlib.util.spec_from_file_location("ref_tokenizer", REF / "music2midi" / "tokenizer.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def embeds(B, S, d, seed=7):
    return torch.from_numpy(synth.normal(seed, "embeds", (B, S, d), 3.0))


# ------------------------------------------------------------------ T5 goldens
def t5_case(name, t5cfg, B, S, L, Ld, eos, out):
    geom = T5Geometry(t5cfg)
    sd = synth.t5_state_dict(geom, seed=0)
    synth.perturb_layer_norms(sd, 0)
    if eos:
        synth.force_eos_head(sd, geom)
    hf = build_hf(t5cfg, sd)
    orc = T5Oracle(geom, sd)
    x = embeds(B, S, geom.d_model)
    with torch.no_grad():
        enc = hf.encoder(inputs_embeds=x).last_hidden_state
        ids = hf.generate(inputs_embeds=x, max_length=L, do_sample=False)
        labels = torch.from_numpy((synth.uniform01(3, "labels", B * Ld) * 330).astype(np.int64).reshape(B, Ld)) + 3
        labels[0, Ld - 3:] = -100   # exercise ignore_index
        fw = hf(inputs_embeds=x, labels=labels)
    # pin the oracle against HF right here
    o_enc = orc.encode(x)
    o_ids, margins = orc.generate(x, L, return_margins=True)
    o_loss, o_logits = orc.forward(x, labels)
    assert torch.equal(ids, o_ids), f"{name}: oracle greedy ids differ from HF"
    assert (enc - o_enc).abs().max() < 1e-4
    assert (fw.logits - o_logits).abs().max() < 2e-3 and abs(fw.loss.item() - o_loss.item()) < 1e-4
    print(f"[{name}] ids {tuple(ids.shape)} min margin {margins.min():.4f} | enc diff {(enc - o_enc).abs().max():.2e} "
          f"| logits diff {(fw.logits - o_logits).abs().max():.2e}")
    rows = sorted({0, 1, S // 2, S - 1})
    out[name] = dict(
        ids=ids.numpy().astype(np.int16), margins=margins.numpy().astype(np.float32),
        enc_rows=np.asarray(rows, dtype=np.int32), enc_sample=enc[:, rows].numpy().astype(np.float32),
        enc_abs_sum=np.float64(enc.double().abs().sum().item()),
        labels=labels.numpy().astype(np.int16), loss=np.float32(fw.loss.item()),
        logits_sample=fw.logits[:, :: max(1, Ld // 4)].numpy().astype(np.float32),
        meta=np.asarray([B, S, L, Ld, int(eos)], dtype=np.int32))


def make_t5(out_path):
    cfg = ref_config()
    full, tiny = dict(cfg["model"]["t5"]), tiny_t5(cfg)
    cases = {}
    t5_case("tiny", tiny, 3, 19, 40, 12, False, cases)
    t5_case("tiny_eos", tiny, 5, 30, 64, 8, True, cases)
    t5_case("full_s190", full, 2, 190, 128, 24, False, cases)
    t5_case("full_eos", full, 4, 60, 96, 8, True, cases)
    t5_case("full_s864", full, 2, 864, 1024, 8, False, cases)
    flat = {f"{c}/{k}": v for c, d in cases.items() for k, v in d.items()}
    np.savez_compressed(out_path, **flat)


# ------------------------------------------------------------------ bf16-mode goldens at the headline size
def make_t5_bf16(out_path):
    """The bf16 throughput mode's pin at BASELINE configs[2]'s geometry (S = 864, max_length 1024): greedy ids and top-2 logit
    margins of ``T5Oracle(emulate="bf16")`` — the oracle that rounds to bfloat16 where the device's bf16 mode stores bfloat16 —
    (a) on the two embedding clips of the ``full_s864`` case (whose fp32 ids are pinned to HuggingFace in t5.npz), and
    (b) on clips 0 and 1 of bench.py's own workload (synthetic 10 s waveforms -> log-mel -> conditioning rows, plain seed-0
    weights), with the fp32 oracle's ids beside them.  The fp32 oracle is the one pinned to HF (make_t5); the bf16 emulation is
    the same code with its rounding points switched on, so this fixture pins the DEVICE against the emulation, not the emulation
    against a third party (none exists for this rounding scheme)."""
    from oracle.logmel import conditioning
    cfg = ref_config()
    geom = T5Geometry(dict(cfg["model"]["t5"]))
    data = {}
    # (a)
    sd = synth.t5_state_dict(geom, seed=0)
    synth.perturb_layer_norms(sd, 0)
    x = embeds(2, 864, geom.d_model)
    ids, margins = T5Oracle(geom, sd, emulate="bf16").generate(x, 1024, return_margins=True)
    ids32 = T5Oracle(geom, sd).generate(x, 1024)
    z = np.load(HERE / "t5.npz")
    assert np.array_equal(ids32.numpy(), z["full_s864/ids"].astype(np.int64)), "the fp32 oracle no longer reproduces the HF golden"
    data["full_s864_bf16/ids"] = ids.numpy().astype(np.int16)
    data["full_s864_bf16/margins"] = margins.numpy().astype(np.float32)
    print(f"[t5_bf16] full_s864: ids {tuple(ids.shape)}, min margin {margins.min():.4f}, agreement with fp32 {(ids == ids32).float().mean():.3f}")
    # (b)
    sd = synth.t5_state_dict(geom, seed=0)
    sp = cfg["spectrogram"]
    T = 220500
    wav = torch.from_numpy(synth.waveform_batch(0, 2, T))
    idx = torch.from_numpy(synth.cond_index_batch(0, 2))
    emb = [torch.from_numpy(sd[f"conditioning.embeds.{i}.weight"]) for i in range(2)]
    xw = conditioning(LogMelOracle(cfg["model"]["sample_rate"], sp["n_fft"], sp["hop_length"], sp["f_min"], geom.d_model)(wav), idx, emb)
    assert xw.shape == (2, 864, geom.d_model)
    ids, margins = T5Oracle(geom, sd, emulate="bf16").generate(xw, 1024, return_margins=True)
    ids32, margins32 = T5Oracle(geom, sd).generate(xw, 1024, return_margins=True)
    data["bench_clips_bf16/ids"] = ids.numpy().astype(np.int16)
    data["bench_clips_bf16/margins"] = margins.numpy().astype(np.float32)
    data["bench_clips_fp32/ids"] = ids32.numpy().astype(np.int16)
    data["bench_clips_fp32/margins"] = margins32.numpy().astype(np.float32)
    print(f"[t5_bf16] bench clips 0-1: min margin bf16 {margins.min():.4f} / fp32 {margins32.min():.4f}, agreement {(ids == ids32).float().mean():.3f}")
    np.savez_compressed(out_path, **data)


# ------------------------------------------------------------------ forced-decode goldens: every position of the headline sequence
NOISE_SEEDS = (123, 124, 125, 126, 127)


def _forced_errs(lg, data, name):
    """|logits - fixture| on the fixture's samples (top-4 of every position + all columns on the stride) and |d (top-1 - top-2)|."""
    lg = lg.numpy().astype(np.float64)
    steps = data[f"{name}/full_steps"].astype(np.int64)
    tv, ti = data[f"{name}/top_vals"].astype(np.float64), data[f"{name}/top_idx"].astype(np.int64)
    e = np.abs(np.take_along_axis(lg, ti, 2) - tv)
    ef = np.abs(lg[:, steps] - data[f"{name}/full_logits"])
    allv = np.concatenate([e.ravel(), ef.ravel()])
    pm = np.take_along_axis(lg, ti[:, :, :2], 2)
    dm = np.abs((pm[:, :, 0] - pm[:, :, 1]) - (tv[:, :, 0] - tv[:, :, 1]))
    return (np.asarray([allv.max(), np.quantile(allv, 0.999), allv.mean()]), np.asarray([dm.max(), np.quantile(dm, 0.999), dm.mean()]))


def forced_noise_floor(geom, name, sd, x, data) -> dict:
    """The bf16 emulation's OWN noise floor along a forced sequence: the same forced pass with the inputs perturbed by 1e-6 (relative).
    bfloat16 rounding decisions flip under any change of summation order, and the flips compound over 12 layers; two evaluations
    that are not bit-identical in every GEMM input — device and emulation sum in different orders — cannot agree better than the
    emulation agrees with itself here.  Stored per perturbation seed (one draw is one sample of a maximum): max / 99.9th
    percentile / mean of |d logit| on the fixture's samples and of |d (top-1 - top-2)|, the quantity an arg-max flip depends on;
    `self_noise_*` = the MEDIAN over the seeds of each statistic — what tests/forced_check.py scales its bars from."""
    ids = torch.from_numpy(data[f"{name}/ids"].astype(np.int64))
    lo, mg = [], []
    for seed in NOISE_SEEDS:
        noise = torch.from_numpy(synth.normal(seed, "noise", tuple(x.shape), 1.0))
        _, lg = T5Oracle(geom, sd, emulate="bf16").forward(x * (1 + 1e-6 * noise), ids[:, 1:].clone())
        a, b = _forced_errs(lg, data, name)
        lo.append(a); mg.append(b)
        print(f"[t5_forced] {name}: self-noise floor, seed {seed}: |d logit| max / p99.9 / mean {a}, |d margin| {b}", flush=True)
    lo, mg = np.stack(lo), np.stack(mg)
    return {f"{name}/self_noise_logit_seeds": lo, f"{name}/self_noise_margin_seeds": mg,
            f"{name}/self_noise_logit": np.median(lo, 0), f"{name}/self_noise_margin": np.median(mg, 0)}


def forced_bf16_vs_fp32(geom, name, sd, x, data) -> dict:
    """What rounding to bfloat16 costs against the fp32 reference itself, on the oracle side: the bf16 EMULATION forced along the fp32
    oracle's ids against the fp32 oracle's logits (max / p99.9 / mean on the fixture's samples, and of the top-2 margin).  The GPU
    test holds the DEVICE's bf16 mode, forced along the same fp32 ids, to a fixed absolute bound next to this figure
    (tests/forced_check.py BF16_VS_FP32_ABS) — a bar that does not come from the emulation's noise."""
    ids = torch.from_numpy(data[f"{name}/ids"].astype(np.int64))
    _, lg = T5Oracle(geom, sd, emulate="bf16").forward(x, ids[:, 1:].clone())
    a, b = _forced_errs(lg, data, name)
    print(f"[t5_forced] {name}: bf16 emulation vs the fp32 oracle along the fp32 ids: |d logit| max / p99.9 / mean {a}, |d margin| {b}", flush=True)
    return {f"{name}/bf16_emulation_err": a, f"{name}/bf16_emulation_margin_err": b}


def forced_cases(geom, cfg):
    """(name stem, state dict, encoder inputs) of the three forced cases."""
    from oracle.logmel import conditioning
    sd = synth.t5_state_dict(geom, seed=0)
    synth.perturb_layer_norms(sd, 0)
    yield "full_s864", sd, embeds(2, 864, geom.d_model)
    sd = synth.t5_state_dict(geom, seed=0)
    sp = cfg["spectrogram"]
    wav = torch.from_numpy(synth.waveform_batch(0, 2, 220500))
    idx = torch.from_numpy(synth.cond_index_batch(0, 2))
    emb = [torch.from_numpy(sd[f"conditioning.embeds.{i}.weight"]) for i in range(2)]
    yield "bench_clips", sd, conditioning(LogMelOracle(cfg["model"]["sample_rate"], sp["n_fft"], sp["hop_length"], sp["f_min"], geom.d_model)(wav), idx, emb)
    sd = synth.t5_state_dict(geom, seed=0)
    synth.perturb_layer_norms(sd, 0)
    yield "native_s190", sd, embeds(2, 190, geom.d_model)


def make_t5_forced_noise(out_path):
    """Recompute only the noise records of an existing t5_forced.npz (the floors over NOISE_SEEDS, the emulation-vs-fp32 figures) from
    the ids it stores — the same helpers make_t5_forced calls, without repeating the six 1 023-step greedy decodes."""
    cfg = ref_config()
    geom = T5Geometry(dict(cfg["model"]["t5"]))
    data = dict(np.load(out_path))
    for stem, sd, x in forced_cases(geom, cfg):
        data.update(forced_noise_floor(geom, f"{stem}_bf16", sd, x, data))
        data.update(forced_bf16_vs_fp32(geom, f"{stem}_fp32", sd, x, data))
    np.savez_compressed(out_path, **data)


def make_t5_forced(out_path):
    """Teacher-forced pin of BOTH precision modes along the WHOLE headline sequence (S = 864, 1 023 positions; ref
    music2midi/transformer.py:41-45).  After a greedy divergence two decodes are different sequences and nothing further compares;
    forcing the ORACLE's own ids through the device's KV-cached decode kernels compares every position regardless.  The greedy
    trajectory's per-step logits ARE the teacher-forced logits along its own ids, so one ``generate`` per case yields: ids, top-2
    margins, the top-4 (value, column) of every step, and the full 400-column logits on a stride of 64 steps.  Cases: the two
    HF-pinned ``full_s864`` embedding clips and clips 0-1 of bench.py's workload, each under the fp32 oracle (pinned to HF by
    make_t5) and under ``emulate="bf16"``; ids are asserted equal to the committed t5.npz / t5_bf16.npz."""
    from oracle.logmel import conditioning
    cfg = ref_config()
    geom = T5Geometry(dict(cfg["model"]["t5"]))
    L, stride = 1024, 64
    zb, z32 = np.load(HERE / "t5_bf16.npz"), np.load(HERE / "t5.npz")
    data = {}

    def run(name, sd, x, emulate, want_ids):
        steps = sorted(set(range(0, L - 1, stride)) | {L - 2})
        top_v, top_i, full = [], [], []

        def hook(t, logits):
            v, i = torch.topk(logits, 4, dim=-1)
            top_v.append(v.clone()); top_i.append(i.clone())
            if t in steps:
                full.append(logits.clone())

        ids, margins = T5Oracle(geom, sd, emulate=emulate).generate(x, L, return_margins=True, logits_hook=hook)
        assert ids.shape == (2, L) and (want_ids is None or np.array_equal(ids.numpy(), np.asarray(want_ids).astype(np.int64))), \
            f"{name}: ids differ from the committed fixture / from HuggingFace"
        data[f"{name}/ids"] = ids.numpy().astype(np.int16)
        data[f"{name}/margins"] = margins.numpy().astype(np.float32)
        data[f"{name}/top_vals"] = torch.stack(top_v, 1).numpy().astype(np.float32)          # [2, 1023, 4]
        data[f"{name}/top_idx"] = torch.stack(top_i, 1).numpy().astype(np.int16)
        data[f"{name}/full_steps"] = np.asarray(steps, dtype=np.int32)
        data[f"{name}/full_logits"] = torch.stack(full, 1).numpy().astype(np.float32)       # [2, n, 400]
        print(f"[t5_forced] {name}: min margin {margins.min():.4f}, positions with margin < 0.5: {(margins < 0.5).sum().item()} of {margins.numel()}, "
              f"|logit| max {torch.stack(top_v, 1).abs().max():.1f}")
        if emulate == "bf16":
            data.update(forced_noise_floor(geom, name, sd, x, data))
        else:
            data.update(forced_bf16_vs_fp32(geom, name, sd, x, data))

    sd = synth.t5_state_dict(geom, seed=0)
    synth.perturb_layer_norms(sd, 0)
    x = embeds(2, 864, geom.d_model)
    run("full_s864_fp32", sd, x, "fp32", z32["full_s864/ids"])
    run("full_s864_bf16", sd, x, "bf16", zb["full_s864_bf16/ids"])
    sd = synth.t5_state_dict(geom, seed=0)
    sp = cfg["spectrogram"]
    wav = torch.from_numpy(synth.waveform_batch(0, 2, 220500))
    idx = torch.from_numpy(synth.cond_index_batch(0, 2))
    emb = [torch.from_numpy(sd[f"conditioning.embeds.{i}.weight"]) for i in range(2)]
    xw = conditioning(LogMelOracle(cfg["model"]["sample_rate"], sp["n_fft"], sp["hop_length"], sp["f_min"], geom.d_model)(wav), idx, emb)
    run("bench_clips_fp32", sd, xw, "fp32", zb["bench_clips_fp32/ids"])
    run("bench_clips_bf16", sd, xw, "bf16", zb["bench_clips_bf16/ids"])
    # The reference's OWN inference geometry (ref config.yaml:16,46-47, model.py:115-134: 3 s segments at 16 kHz -> S = 190,
    # max_length 1024): the two clips of the `full_s190` case, decoded to the full 1 024 tokens by HuggingFace itself here — the
    # fp32 oracle's ids are asserted equal to HF's over all 1 024 — and the bf16 emulation beside it.
    sd = synth.t5_state_dict(geom, seed=0)
    synth.perturb_layer_norms(sd, 0)
    x190 = embeds(2, 190, geom.d_model)
    with torch.no_grad():
        ids_hf = build_hf(dict(cfg["model"]["t5"]), sd).generate(inputs_embeds=x190, max_length=L, do_sample=False)
    assert ids_hf.shape == (2, L) and np.array_equal(ids_hf[:, :128].numpy(), z32["full_s190/ids"].astype(np.int64))
    run("native_s190_fp32", sd, x190, "fp32", ids_hf.numpy())
    run("native_s190_bf16", sd, x190, "bf16", None)
    np.savez_compressed(out_path, **data)


# ------------------------------------------------------------------ frontend goldens
def make_frontend(out_path):
    cfg = ref_config()
    sr, n_mels = cfg["model"]["sample_rate"], cfg["model"]["t5"]["d_model"]
    sp = cfg["spectrogram"]
    fb = melscale_fbanks(sp["n_fft"] // 2 + 1, sp["f_min"], float(sr // 2), n_mels, sr).numpy()
    r, c = np.nonzero(fb)
    orc = LogMelOracle(sr, sp["n_fft"], sp["hop_length"], sp["f_min"], n_mels)
    data = dict(fb_rows=r.astype(np.int16), fb_cols=c.astype(np.int16), fb_vals=fb[r, c].astype(np.float32),
                fb_shape=np.asarray(fb.shape, dtype=np.int32))
    for kind in ("noise", "tones", "zeros", "music"):
        wav = torch.from_numpy(synth.waveform_batch(0, 2, 4096, kind))
        data[f"logmel_{kind}"] = orc(wav).numpy().astype(np.float32)             # [2, 17, 384]
        data[f"logmel64_{kind}"] = orc(wav, dtype=torch.float64).numpy().astype(np.float32)
    wav = torch.from_numpy(synth.waveform_batch(5, 1, 48000, "noise"))
    full = orc(wav)
    data["logmel_noise_48000_frames"] = np.asarray([0, 1, 94, 187], dtype=np.int32)
    data["logmel_noise_48000"] = full[:, [0, 1, 94, 187]].numpy().astype(np.float32)
    # music-like material (decaying harmonic notes + digital silence) at the reference-native and the
    # BASELINE clip length: float32 torch.stft path AND the float64 evaluation of the same formula, on
    # sampled frames (start, onset region, middle, the silent tail, last frame)
    for T in (48000, 220500):
        wav = torch.from_numpy(synth.waveform_batch(3, 1, T, "music"))
        F = 1 + T // 256
        frames = sorted({0, 1, F // 7, F // 3, F // 2, (2 * F) // 3, int(F * 0.85) - 4, int(F * 0.85) + 9, F - 1})
        data[f"logmel_music_{T}_frames"] = np.asarray(frames, dtype=np.int32)
        data[f"logmel_music_{T}"] = orc(wav)[:, frames].numpy().astype(np.float32)
        data[f"logmel64_music_{T}"] = orc(wav, dtype=torch.float64)[:, frames].numpy()          # float64 kept: the truth
    print(f"[frontend] fb nnz {len(r)} taps/filter min {np.bincount(c).min()} max {np.bincount(c).max()}")
    np.savez_compressed(out_path, **data)


# ------------------------------------------------------------------ tokenizer goldens
def random_notes(seed, n, max_t=9.5):
    u = synth.uniform01(seed, "notes", n * 4).reshape(n, 4)
    onset = np.sort(u[:, 0] * max_t)
    dur = u[:, 1] * 1.2
    dur[u[:, 3] < 0.15] = 0.0          # zero-length notes -> min one step
    pitch = np.floor(21 + u[:, 2] * 88)
    return np.stack([onset, onset + dur, pitch, np.full(n, 80.0)], axis=1)


def make_tokenizer(out_path):
    ref = import_reference_tokenizer()
    cfg = ref_config()
    tok = ref.MidiTokenizer(ConfigNode(cfg))
    cases = []
    note_sets = {
        "empty": np.zeros((0, 4)),
        "single": np.array([[0.0, 0.5, 60, 80]]),
        "chord": np.array([[0.0, 0.5, 60, 80], [0.0, 0.5, 64, 80], [0.0, 1.0, 67, 80]]),
        "tie_half_step": np.array([[0.025, 0.075, 60, 80], [0.125, 0.175, 61, 80]]),
        "clip_late": np.array([[9.9, 10.5, 72, 80], [11.0, 12.0, 73, 80]]),
        "zero_len": np.array([[1.0, 1.0, 50, 80], [1.0, 0.9, 51, 80]]),
        "rand12": random_notes(1, 12), "rand40": random_notes(2, 40), "rand90": random_notes(3, 90, 2.9),
    }
    for name, notes in note_sets.items():
        for cutoff in (None, 3):
            ids = tok._tokenize(notes, cutoff).numpy().tolist()
            cases.append(dict(kind="encode", name=name, cutoff=cutoff, notes=notes.tolist(), ids=ids))
    batch = [note_sets["rand12"], note_sets["empty"], note_sets["chord"]]
    cases.append(dict(kind="encode_batch", notes=[b.tolist() for b in batch], ids=tok(batch).numpy().tolist()))
    # decode vectors: valid streams, streams with garbage, unmatched onsets, ids in the unused 333..399 range
    streams = [tok._tokenize(note_sets[k]).numpy() for k in ("rand12", "rand40", "chord", "zero_len", "empty")]
    rnd = (synth.uniform01(9, "stream", 3 * 200).reshape(3, 200) * 400).astype(np.int64)
    rnd[0, 150] = 2
    streams += [rnd[0], rnd[1], rnd[2], np.array([1, 133, 3, 65, 66, 140, 4, 65, 0, 0, 2, 150, 3, 70])]
    for i, s in enumerate(streams):
        for cutoff in (None, 3):
            notes = tok._decode(np.array(s), 0, cutoff)
            cases.append(dict(kind="decode", name=f"s{i}", cutoff=cutoff, ids=np.asarray(s).tolist(), notes=notes.tolist()))
    seq = tok.decode([streams[0], streams[2], streams[5]], mode="sequential", duration_per_batch=3)
    cases.append(dict(kind="decode_sequential", duration=3, ids=[np.asarray(s).tolist() for s in (streams[0], streams[2], streams[5])],
                      notes=seq.tolist()))
    bat = tok.decode(torch.tensor(tok(batch)), mode="batched")
    cases.append(dict(kind="decode_batched", ids=tok(batch).numpy().tolist(), notes=[b.tolist() for b in bat]))
    cases.append(dict(kind="to_string", ids=[0, 1, 2, 3, 4, 5, 132, 133, 332], names=tok.to_string(np.array([0, 1, 2, 3, 4, 5, 132, 133, 332]))))
    out_path.write_text(json.dumps(cases))
    print(f"[tokenizer] {len(cases)} cases")

# ------------------------------------------------------------------ training goldens
def make_train(out_path):
    """HF T5ForConditionalGeneration (eager, untied head) loss + autograd gradients, and three steps of HF's own
    transformers.optimization.Adafactor(warmup_init=True), on the tiny config: the pin of oracle/train.py."""
    from transformers.optimization import Adafactor
    from oracle.train import AdafactorOracle, T5TrainOracle, leaf_params
    cfg = ref_config()
    t5cfg = tiny_t5(cfg)
    geom = T5Geometry(t5cfg)
    sd = synth.t5_state_dict(geom, seed=0)
    synth.perturb_layer_norms(sd, 0)
    B, F, Ld = 3, 21, 14
    feats = torch.from_numpy(synth.normal(5, "feats", (B, F, geom.d_model), 2.0))
    cond = torch.from_numpy(synth.cond_index_batch(2, B))
    labels = torch.from_numpy((synth.uniform01(4, "labels", B * Ld) * 330).astype(np.int64).reshape(B, Ld)) + 3
    labels[1, Ld - 4:] = -100
    labels[2, Ld - 1:] = -100
    hf = build_hf(t5cfg, sd)          # eval(): dropout off - the deterministic part of the step is what can be pinned
    emb = [torch.nn.Parameter(torch.from_numpy(sd[f"conditioning.embeds.{i}.weight"]).clone()) for i in range(2)]
    hf_params = list(hf.parameters()) + emb
    opt = Adafactor(hf_params, warmup_init=True)

    def hf_loss():
        x = torch.cat([torch.stack([emb[i][cond[:, i]] for i in range(2)], dim=1), feats], dim=1)
        return hf(inputs_embeds=x, labels=labels)

    orc_params = leaf_params(sd)
    orc = T5TrainOracle(geom, orc_params)
    oopt = AdafactorOracle(orc_params)
    names = {"transformer." + k: v for k, v in hf.named_parameters()}
    names.update({f"conditioning.embeds.{i}.weight": emb[i] for i in range(2)})
    assert set(names) == set(orc_params), sorted(set(names) ^ set(orc_params))[:5]
    data = dict(meta=np.asarray([B, F, Ld], dtype=np.int32), labels=labels.numpy().astype(np.int16))
    keys = sorted(orc_params)
    data["keys"] = np.asarray(keys)
    for step in range(3):
        opt.zero_grad()
        out = hf_loss()
        out.loss.backward()
        loss_o, logits_o, grads_o = orc.loss_and_grads(feats, cond, labels)
        assert abs(out.loss.item() - loss_o.item()) < 2e-5, (out.loss.item(), loss_o.item())
        worst = 0.0
        for k in keys:
            gh, go = names[k].grad, grads_o[k]
            worst = max(worst, float((gh - go).abs().max() / (gh.abs().max() + 1e-12)))
        assert worst < 2e-4, worst
        if step == 0:
            assert (out.logits - logits_o).abs().max() < 2e-3
            data["logits_sample"] = out.logits[:, ::4].detach().numpy().astype(np.float32)
            data["grad_abs_sum"] = np.asarray([names[k].grad.double().abs().sum().item() for k in keys])
            data["grad_l2"] = np.asarray([names[k].grad.double().norm().item() for k in keys])
            data["grad_head"] = np.stack([np.resize(names[k].grad.reshape(-1)[:8].numpy(), 8) for k in keys]).astype(np.float32)
        data.setdefault("losses", []).append(out.loss.item())
        opt.step()
        oopt.step(grads_o)
        drift = max(float((names[k].detach() - orc_params[k].detach()).abs().max()) for k in keys)
        assert drift < 2e-6, (step, drift)
        print(f"[train] step {step}: loss {out.loss.item():.6f} | oracle grad rel diff {worst:.2e} | param drift vs HF Adafactor {drift:.2e}")
    data["losses"] = np.asarray(data["losses"], dtype=np.float64)
    data["param_abs_sum_after3"] = np.asarray([names[k].detach().double().abs().sum().item() for k in keys])
    data["param_head_after3"] = np.stack([np.resize(names[k].detach().reshape(-1)[:8].numpy(), 8) for k in keys]).astype(np.float32)
    np.savez_compressed(out_path, **data)


if __name__ == "__main__":
    torch.manual_seed(0)
    only = set(sys.argv[1:])            # e.g. `make_golden.py frontend` regenerates one fixture
    if not only or "tokenizer" in only:
        make_tokenizer(HERE / "tokenizer_cases.json")
    if not only or "frontend" in only:
        make_frontend(HERE / "frontend.npz")
    if not only or "t5" in only:
        make_t5(HERE / "t5.npz")
    if not only or "train" in only:
        make_train(HERE / "train.npz")
    if not only or "t5_bf16" in only:
        make_t5_bf16(HERE / "t5_bf16.npz")
    if not only or "t5_forced" in only:
        make_t5_forced(HERE / "t5_forced.npz")
    if "t5_forced_noise" in only:
        make_t5_forced_noise(HERE / "t5_forced.npz")
    for p in sorted(HERE.glob("*.npz")) + sorted(HERE.glob("*.json")):
        print(p.name, p.stat().st_size, "bytes")
    # never leave bytecode in the read-only reference tree
    assert not (REF / "music2midi" / "__pycache__").exists()
