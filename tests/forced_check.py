"""Teacher-forced check of a precision mode along EVERY position of the headline sequence (checker code: used by
tests/test_golden_gpu.py and by bench.py's `parity_mode` record; reads tests/golden/t5_forced.npz, never the oracle).

ref: music2midi/transformer.py:41-45 -> HF greedy decode.  After a first greedy divergence device and oracle are different
sequences; feeding the ORACLE's ids to the device's KV-cached decode kernels (`m2m_decode_forced` with M2M_FORWARD=step: the
same dec_attn / dec_ff / lm_head kernels the greedy loop launches, head kernel in forced mode) makes all 1 023 positions
comparable: the arg-max wherever the oracle's top-2 margin exceeds twice the logit error bound, and the logits themselves on
the fixture's samples (top-4 columns of every position + all 400 columns on a stride of 64 positions)."""
from __future__ import annotations

import os
from pathlib import Path

import numpy as np
import torch

FIXTURE = Path(__file__).resolve().parent / "golden" / "t5_forced.npz"

# fp32: |device logit - oracle logit| <= 2e-3 on every sampled entry (the bar every fp32 logits test of this repo uses) and the same
# arg-max wherever the oracle's top-2 margin exceeds twice that.
# bf16: held against the EMULATION'S OWN NOISE FLOOR, stored in the fixture (make_golden.py: the same forced pass of the bf16-emulating
# oracle with its inputs perturbed by 1e-6 — bfloat16 rounding decisions flip under any change of summation order and compound over
# twelve layers, so two evaluations that sum in different orders cannot agree better than that).  Measured on MI355X (round 4): the
# device's errors ARE that floor (max / 99.9th percentile / mean 0.411 / 0.333 / 0.0781 against 0.488 / 0.342 / 0.0789 on full_s864).
# Bars: maximum <= 1.5 x, 99.9th percentile <= 1.3 x, mean <= 1.2 x the floor's; arg-max equal wherever the oracle's margin exceeds
# 1.5 x the floor's largest margin change.  This replaces round 3's reasoned margin of 0.5.
# Round 5: the floor is the MEDIAN over five perturbation seeds (one draw is one sample of a maximum; the per-seed values
# are in the fixture), the arg-max threshold is capped at 0.5 — the margin round 3 reasoned, never looser — and the bf16 mode is
# ALSO held to bars that do not come from the emulation at all: the device's bf16 logits, forced along the FP32 oracle's ids, against
# the fp32 oracle's logits (`forced_bf16_vs_fp32`, BF16_VS_FP32_ABS: fixed absolute numbers on logits that reach 87; the emulation
# itself sits at 0.62 / 0.48 / 0.114 there, fixture `bf16_emulation_err`).  A defect the size of the emulation's noise in ONE place
# can hide under a noise-scaled bar; it cannot also keep the distance to the fp32 reference inside a fixed one.
FP32_LOGIT_ERR_BOUND = 2e-3
BF16_FLOOR_FACTORS = (1.5, 1.3, 1.2)          # on (max, 99.9th percentile, mean) of the stored self-noise
BF16_ARGMAX_MARGIN_CAP = 0.5
BF16_VS_FP32_ABS = (1.0, 0.70, 0.16)          # |device bf16 logit - fp32 oracle logit|: max, 99.9th percentile, mean


def bf16_margin_threshold(z, case: str) -> float:
    """Oracle top-2 margin above which the device's bf16 arg-max must equal the emulation's: 1.5 x the emulation's own largest margin
    change (median over the perturbation seeds), never more than 0.5."""
    return float(min(BF16_FLOOR_FACTORS[0] * float(z[f"{case}/self_noise_margin"][0]), BF16_ARGMAX_MARGIN_CAP))


def forced_logits(model, x: torch.Tensor, dec_in: torch.Tensor, mode: str) -> np.ndarray:
    """mode "step": Ld KV-cached decode steps (the decode kernels); "batched": the one-pass teacher-forced decoder."""
    old = os.environ.get("M2M_FORWARD")
    try:
        if mode == "step":
            os.environ["M2M_FORWARD"] = "step"
        else:
            os.environ.pop("M2M_FORWARD", None)
        return model.logits_from_embeds(x, dec_in).cpu().numpy()
    finally:
        if old is None:
            os.environ.pop("M2M_FORWARD", None)
        else:
            os.environ["M2M_FORWARD"] = old


def forced_check(model, x2: torch.Tensor, case: str, precision: str, copies: int = 16, mode: str = "step", z=None) -> dict:
    """x2: the case's two encoder-input clips [2, S, d] (device).  The two clips are tiled `copies` times into one batch (32 clips at
    copies = 16: the headline batch, so the K/V stream takes the non-temporal path it takes in the benchmark) and every copy must
    produce bit-identical logits (batch invariance).  Returns the record; raises AssertionError with the position on a violation."""
    z = np.load(FIXTURE) if z is None else z
    ids = z[f"{case}/ids"].astype(np.int64)                    # [2, 1024]: start token + 1 023 greedy tokens of the oracle (pad after a row's EOS)
    margins = z[f"{case}/margins"].astype(np.float64)          # [2, 1023]
    top_v, top_i = z[f"{case}/top_vals"].astype(np.float64), z[f"{case}/top_idx"].astype(np.int64)
    steps, full = z[f"{case}/full_steps"].astype(np.int64), z[f"{case}/full_logits"].astype(np.float64)
    Ld = ids.shape[1] - 1
    dec_in = torch.from_numpy(ids[:, :Ld]).repeat(copies, 1)
    xx = x2.repeat(copies, 1, 1).contiguous()
    logits = forced_logits(model, xx, dec_in.to(xx.device), mode)          # [2 * copies, Ld, V]
    assert np.isfinite(logits).all()
    for c in range(1, copies):
        assert np.array_equal(logits[2 * c: 2 * c + 2], logits[:2]), f"{case}: copy {c} of the clips differs from copy 0 (batch invariance)"
    dev = logits[:2].astype(np.float64)
    if precision == "fp32":
        bars = (FP32_LOGIT_ERR_BOUND, FP32_LOGIT_ERR_BOUND, FP32_LOGIT_ERR_BOUND)
        margin_thr = 2 * FP32_LOGIT_ERR_BOUND
    else:
        floor, floor_m = z[f"{case}/self_noise_logit"], z[f"{case}/self_noise_margin"]
        bars = tuple(float(f * v) for f, v in zip(BF16_FLOOR_FACTORS, floor))
        margin_thr = bf16_margin_threshold(z, case)
    err_top = np.abs(np.take_along_axis(dev, top_i, axis=2) - top_v)      # [2, Ld, 4]
    err_full = np.abs(dev[:, steps] - full)                                # [2, n, V]
    errs = np.concatenate([err_top.ravel(), err_full.ravel()])
    am = dev.argmax(-1)
    want = top_i[:, :, 0]        # the oracle's arg-max (= its next id while the row is live; after a row's EOS the ids are pad, the logits go on)
    agree = am == want
    checked = margins > margin_thr
    bad = checked & ~agree
    rec = {"case": case, "mode": mode, "positions": int(agree.size), "argmax_agree": int(agree.sum()),
           "argmax_asserted_positions": int(checked.sum()), "margin_threshold": margin_thr,
           "logit_err_bars_max_p999_mean": list(bars),
           "max_logit_err": float(errs.max()), "p999_logit_err": float(np.quantile(errs, 0.999)), "mean_logit_err": float(errs.mean()),
           "max_logit_err_late_half": float(max(err_top[:, Ld // 2:].max(), err_full[:, steps >= Ld // 2].max())),
           "logit_scale": float(np.abs(top_v).max()), "logit_samples": int(errs.size),
           "smallest_margin_with_agreement": float(margins[agree].min()),
           "largest_margin_with_disagreement": float(margins[~agree].max()) if (~agree).any() else 0.0}
    if bad.any():
        b, t = np.argwhere(bad)[0]
        raise AssertionError(f"{case} [{precision}, {mode}]: position {t} of row {b}: device arg-max {am[b, t]} != oracle id {want[b, t]} "
                             f"at oracle margin {margins[b, t]:.4f} (> {margin_thr:.4g}); record {rec}")
    got = (rec["max_logit_err"], rec["p999_logit_err"], rec["mean_logit_err"])
    assert all(g <= b for g, b in zip(got, bars)), f"{case} [{precision}, {mode}]: |device - oracle| logit error max / p99.9 / mean {got} above {bars}; record {rec}"
    return rec


def forced_bf16_vs_fp32(model_bf16, x2: torch.Tensor, case_fp32: str, copies: int = 16, z=None) -> dict:
    """The bf16 mode against the FP32 reference itself: the fp32 oracle's ids (fixture case `*_fp32`, pinned to HuggingFace) forced
    through the bf16 decode kernels, |device logit - fp32 oracle logit| on the fixture's samples held to the fixed BF16_VS_FP32_ABS,
    and the arg-max equal wherever the fp32 oracle's margin exceeds twice the measured maximum error."""
    z = np.load(FIXTURE) if z is None else z
    ids = z[f"{case_fp32}/ids"].astype(np.int64)
    margins = z[f"{case_fp32}/margins"].astype(np.float64)
    top_v, top_i = z[f"{case_fp32}/top_vals"].astype(np.float64), z[f"{case_fp32}/top_idx"].astype(np.int64)
    steps, full = z[f"{case_fp32}/full_steps"].astype(np.int64), z[f"{case_fp32}/full_logits"].astype(np.float64)
    Ld = ids.shape[1] - 1
    dec_in = torch.from_numpy(ids[:, :Ld]).repeat(copies, 1)
    xx = x2.repeat(copies, 1, 1).contiguous()
    dev = forced_logits(model_bf16, xx, dec_in.to(xx.device), "step")[:2].astype(np.float64)
    errs = np.concatenate([np.abs(np.take_along_axis(dev, top_i, axis=2) - top_v).ravel(), np.abs(dev[:, steps] - full).ravel()])
    got = (float(errs.max()), float(np.quantile(errs, 0.999)), float(errs.mean()))
    agree = dev.argmax(-1) == top_i[:, :, 0]
    checked = margins > 2 * got[0]
    rec = {"case": case_fp32, "positions": int(agree.size), "argmax_agree_with_fp32_oracle": int(agree.sum()),
           "argmax_asserted_positions": int(checked.sum()), "max_p999_mean_logit_err_vs_fp32_oracle": list(got),
           "bars": list(BF16_VS_FP32_ABS), "emulation_err_vs_fp32_oracle": [float(v) for v in z[f"{case_fp32}/bf16_emulation_err"]]
           if f"{case_fp32}/bf16_emulation_err" in z else None, "logit_scale": float(np.abs(top_v).max())}
    assert all(g <= b for g, b in zip(got, BF16_VS_FP32_ABS)), f"{case_fp32}: bf16 device vs fp32 oracle logit error max / p99.9 / mean {got} above {BF16_VS_FP32_ABS}; {rec}"
    assert not (checked & ~agree).any(), f"{case_fp32}: bf16 arg-max differs from the fp32 oracle at a margin above twice the measured error; {rec}"
    return rec


def case_inputs(case: str, geom, device):
    """(state dict, encoder inputs [2, 864, d]) of a fixture case, as tests/golden/make_golden.py::make_t5_forced built them.  The bench
    clips go through the ORACLE's log-mel so that the comparison isolates the transformer (device and oracle log-mel differ by
    <= 1e-4, which the frontend tests pin on their own)."""
    from music2midi_amd import synth
    sd = synth.t5_state_dict(geom, seed=0)
    if case.startswith("full_s864") or case.startswith("native_s190"):
        synth.perturb_layer_norms(sd, 0)
        x = torch.from_numpy(synth.normal(7, "embeds", (2, 864 if case.startswith("full_s864") else 190, geom.d_model), 3.0))
    else:
        from oracle.logmel import LogMelOracle, conditioning
        wav = torch.from_numpy(synth.waveform_batch(0, 2, 220500))
        idx = torch.from_numpy(synth.cond_index_batch(0, 2))
        emb = [torch.from_numpy(sd[f"conditioning.embeds.{i}.weight"]) for i in range(2)]
        x = conditioning(LogMelOracle(16000, 2048, 256, 20.0, geom.d_model)(wav), idx, emb)
    return sd, x.to(device)
