#!/usr/bin/env python3
"""Headline benchmark: decoded MIDI tokens / second on synthetic 10 s clips.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision bf16|fp32] [--batch 32]

A "step" is one pass of the whole hot path over one batch of synthetic clips:
fused STFT/log-mel frontend -> conditioning rows -> T5 encoder -> cross-K/V
projection -> 1023 KV-cached greedy decode steps (max_length 1024; random-init
weights never emit EOS, so every clip yields exactly 1023 new tokens).
Workload = BASELINE.json configs[2] (bf16, batch 32 per GPU, 220 500-sample
clips -> encoder length 864); with N > 1 every rank decodes its own 32 clips
(weak scaling, configs[3]) after one RCCL weight broadcast, and the decoded
token matrices are all-gathered every step.  Waveforms are resident in HBM when
the timed region starts.

Prints ONE JSON line (rank 0).  `roofline` is measured live with hipEvents on
the stream the kernels run on (m2m_bench_kernel); `cpu_baseline` times the
oracle (a PyTorch-CPU restatement of the reference path — the reference's own
Python cannot run here: torchaudio/lightning/omegaconf are absent) on a bounded
sample on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured-achievable)
N_SAMPLES = 220500        # 10 s @ 22.05 kHz
MAX_LENGTH = 1024


def cpu_baseline(cfg, state, new_tokens: int, clips: int = 8, threads: int = 16):
    """Oracle on the host cores: `clips` clips, frontend + encoder + `new_tokens` greedy steps, fp32.

    16 intra-op threads: measured fastest on the GPU box's 256-core host (tools/cpu_threads_probe.py:
    4/8/16/32/64/128 threads -> 198/205/211/99/44/18 tokens/s for one clip; the matmuls are tiny)."""
    from music2midi_amd import synth
    from music2midi_amd.config import T5Geometry
    from oracle.logmel import LogMelOracle, conditioning
    from oracle.t5 import T5Oracle

    geom = T5Geometry(cfg.model.t5)
    threads = min(threads, os.cpu_count() or threads)
    torch.set_num_threads(threads)
    fe = LogMelOracle(cfg.model.sample_rate, cfg.spectrogram.n_fft, cfg.spectrogram.hop_length,
                      cfg.spectrogram.f_min, geom.d_model)
    orc = T5Oracle(geom, state, emulate="fp32")
    wav = torch.from_numpy(synth.waveform_batch(0, clips, N_SAMPLES))
    idx = torch.from_numpy(synth.cond_index_batch(0, clips))
    emb = [torch.from_numpy(state[f"conditioning.embeds.{i}.weight"]) for i in range(2)]
    t0 = time.perf_counter()
    x = conditioning(fe(wav), idx, emb)
    ids = orc.generate(x, new_tokens + 1)
    dt = time.perf_counter() - t0
    n = (ids.shape[1] - 1) * clips
    return {"value": n / dt, "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": f"{clips} clips x {N_SAMPLES} samples in one batch: log-mel + encoder (S=864) + {ids.shape[1] - 1} greedy "
                      f"decode steps each, fp32 torch-CPU oracle, {dt:.1f} s wall, {threads} threads of os.cpu_count()={os.cpu_count()}"}


def pmc_traffic_bytes(kernel_substr: str, batch: int):
    """HBM bytes per launch of a kernel from the committed PMC summary (profiles/, collected by
    tools/pmc_traffic.sh in separate FETCH_SIZE / WRITE_SIZE passes with 32-clip launches; FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950).  None when no summary is present."""
    import re
    best = None
    for f in sorted((ROOT / "profiles").glob("*pmc_traffic_summary.txt")):
        vals = []
        for line in f.read_text().splitlines():
            if kernel_substr in line:
                m = re.search(r"x2 corrected\s+([0-9.]+) MB\).*WRITE_SIZE/launch\s+[0-9.]+ KiB \(\s*([0-9.]+) MB\)", line)
                if m:
                    vals.append((float(m.group(1)) + float(m.group(2))) * 1e6 * batch / 32.0)
        if vals:   # several template variants of the kernel (cache policy): mean per launch
            best = sum(vals) / len(vals)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--cpu-tokens", type=int, default=1023, help="greedy steps per clip of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--max-length", type=int, default=MAX_LENGTH,
                    help="decoder max_length (profiling runs only; the headline number uses 1024)")
    args = ap.parse_args()

    from music2midi_amd import distributed as D
    from music2midi_amd import native, synth
    from music2midi_amd.checkpoint import load_t5_state
    from music2midi_amd.config import T5Geometry, default_config
    from music2midi_amd.input import ModelInputs
    from music2midi_amd.transformer import T5Transformer

    rank, local_rank, world = D.init_process_group()
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                  f"--nproc-per-node {args.gpus}", file=sys.stderr)
        args.gpus = world
    native.require_gpu()
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)

    cfg = default_config()
    geom = T5Geometry(cfg.model.t5)
    B = args.batch

    # ---- weights: rank 0 owns them, everyone else receives them over RCCL ----
    model = T5Transformer(cfg.to_dict(), precision=args.precision)
    state = None
    if rank == 0:
        state = synth.t5_state_dict(geom, seed=0)
        load_t5_state(model, state, strict=False)
    model = model.to(dev).eval()
    bcast_bytes = D.broadcast_module_state(model, src=0)

    # ---- synthetic clips of this rank, resident in HBM ----
    first = rank * B
    wav = torch.from_numpy(synth.waveform_batch(first, B, N_SAMPLES)).to(dev)
    cond = torch.from_numpy(synth.cond_index_batch(first, B)).to(dev)
    inputs = ModelInputs(input_waveform=wav, cond_index=cond)

    def step():
        toks = model.generate(inputs, max_length=args.max_length)
        return D.all_gather_tokens(toks, args.max_length, geom.pad_token_id)

    for _ in range(args.warmup):
        step()
    if world > 1:   # communicator set-up must never land in the timed region (even with --warmup 0)
        D.all_gather_tokens(torch.zeros((B, 2), dtype=torch.long, device=dev), args.max_length, geom.pad_token_id)
        D.all_reduce_max(0.0, dev)
        D.all_reduce_sum(0.0, dev)
    D.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        toks = step()
    torch.cuda.synchronize(dev)
    D.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = D.all_reduce_max(elapsed, dev)
    new_tokens_local = (toks.shape[1] - 1) * B
    total_tokens = D.all_reduce_sum(float(new_tokens_local), dev) * args.steps
    assert toks.shape[0] == B * world, toks.shape

    out = None
    if rank == 0:
        value = total_tokens / elapsed
        out = {
            "metric": "decoded MIDI tokens/sec/node on 10 s clips",
            "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: full generate (log-mel + encoder + KV-cached greedy decode), "
                                   f"{args.precision}, batch {B} clips/GPU x {N_SAMPLES} samples, S=864, max_length {args.max_length}"
                                   + (f"; configs[3] sharding over {world} GPUs" if world > 1 else ""),
                       "global_batch": B * world, "clips_per_gpu": B, "new_tokens_per_clip": toks.shape[1] - 1,
                       "parallelism": f"clip-sharded x{world}", "weight_broadcast_bytes": bcast_bytes},
        }

    # ---- phase timings + roofline of the dominant kernel (rank 0 only, N = 1) ----
    if rank == 0 and world == 1 and not args.no_roofline:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        x = model.encoder_inputs(inputs)
        ev[1].record()
        sess, _ = model._encode(x, MAX_LENGTH)
        ev[2].record()
        torch.cuda.synchronize(dev)
        t_dec = time.perf_counter()
        model.generate_from_embeds(x, max_length=MAX_LENGTH)
        torch.cuda.synchronize(dev)
        t_dec = time.perf_counter() - t_dec
        fe_ms, enc_ms = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
        es = 2 if args.precision == "bf16" else 4
        # dominant kernel: decode cross-attention (6 launches per decode step, streams the
        # per-clip cross K/V: 2 * S * inner * esize bytes per clip per launch)
        t_mid = MAX_LENGTH // 2
        cross_us, cross_bytes = model.bench_kernel(native.KERNEL_DEC_CROSS_ATTN, t_mid, 600)
        self_us, self_bytes = model.bench_kernel(native.KERNEL_DEC_SELF_ATTN, t_mid, 600)
        step_us, _ = model.bench_kernel(native.KERNEL_DEC_STEP, t_mid, 200)
        achieved = cross_bytes / (cross_us * 1e-6) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "dec_attn_kernel (cross-attention, decode step)",
                           "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS,
                           "traffic": pmc_traffic_bytes("dec_attn_kernel<m2m::bf16_t, false,", B)
                           if args.precision == "bf16" else None,
                           "algorithmic_bytes_per_launch": cross_bytes, "avg_launch_us": cross_us}
        params_step = 15201664  # decoder weights read once per step, elements
        bytes_step = params_step * es + 6 * (cross_bytes + self_bytes)
        out["extras"] = {
            "frontend_ms": fe_ms, "encoder_plus_crosskv_ms": enc_ms,
            "decode_s": t_dec - (fe_ms + enc_ms) * 0.0, "generate_from_embeds_s": t_dec,
            "self_attn_us_at_t512": self_us, "self_attn_GBs": self_bytes / (self_us * 1e-6) / 1e9,
            "decode_step_us_at_t512": step_us,
            "decode_step_algorithmic_GBs": bytes_step / (step_us * 1e-6) / 1e9,
            "frontend_GBs": B * (4 * N_SAMPLES + 4 * 862 * 384) / (fe_ms * 1e-3) / 1e9,
        }
    if rank == 0 and world == 1 and args.cpu_tokens > 0:
        out["cpu_baseline"] = cpu_baseline(cfg, state, args.cpu_tokens)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
