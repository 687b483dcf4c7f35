#!/usr/bin/env python3
"""RCCL on the box's GPU with a one-rank group: the collectives of the multi-GPU paths (weight broadcast, token all-gather,
gradient all-reduce, the bench's scalar reductions and barrier) run through torch.distributed's nccl backend exactly as
`bench.py --gpus N` issues them, so a broken RCCL / IPC environment shows up on a 1-GPU box and not first on the 8-GPU node."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
flat = torch.arange(30_400_000, dtype=torch.float32, device=dev)            # the 121.6 MB weight / gradient image
t0 = time.perf_counter(); dist.broadcast(flat, src=0); torch.cuda.synchronize(); t_b = time.perf_counter() - t0
t0 = time.perf_counter(); dist.all_reduce(flat, op=dist.ReduceOp.SUM); torch.cuda.synchronize(); t_r = time.perf_counter() - t0
assert float(flat[12345].item()) == 12345.0
send = torch.full((32, 1025), 7, dtype=torch.long, device=dev)
out = torch.empty((32, 1025), dtype=torch.long, device=dev)
dist.all_gather_into_tensor(out, send)
assert torch.equal(out, send)
parts = [torch.empty_like(send)]
dist.all_gather(parts, send)
assert torch.equal(parts[0], send)
v = torch.tensor([3.5], dtype=torch.float64, device=dev)
dist.all_reduce(v, op=dist.ReduceOp.MAX); dist.all_reduce(v, op=dist.ReduceOp.SUM)
assert float(v.item()) == 3.5
# the overlapped gradient averaging of data-parallel training, as distributed.all_reduce_gradients_overlapped issues it: a real
# training pass with a sync stream, the early pieces asynchronously on that stream, the rest on the current one, then the wait
from music2midi_amd import distributed as D, synth
from music2midi_amd.checkpoint import load_t5_state
from music2midi_amd.config import T5Geometry, default_config
from music2midi_amd.training import NativeTrainer
from music2midi_amd.transformer import T5Transformer
import numpy as np
cfg = default_config(); geom = T5Geometry(cfg.model.t5)
model = T5Transformer(cfg.to_dict(), precision="fp32"); load_t5_state(model, synth.t5_state_dict(geom, 0), strict=False); model = model.cuda()
B, S, Ld = 4, 64, 32
tr = NativeTrainer(model, B, S, Ld, precision="bf16")
tr_ref = NativeTrainer(model, B, S, Ld, precision="bf16")
tr.set_sync_stream(torch.cuda.Stream())
x = torch.from_numpy(synth.normal(1, "x", (B, S, 384), 2.0)).cuda(); cond = torch.from_numpy(synth.cond_index_batch(0, B)).cuda()
labels = (torch.from_numpy((synth.uniform01(4, "l", B * Ld) * 330).astype(np.int64).reshape(B, Ld)) + 3).cuda()
early, late = D.split_ranges(tr.n_floats, tr.early_ranges)
for it in range(3):
    tr.forward_backward(x, cond, labels)
    pending = []
    with torch.cuda.stream(tr.sync_stream):
        for o, c in early: pending.append(dist.all_reduce(tr.grads[o:o + c], op=dist.ReduceOp.SUM, async_op=True))
    for o, c in late: dist.all_reduce(tr.grads[o:o + c], op=dist.ReduceOp.SUM)
    for w in pending: w.wait()
    tr_ref.forward_backward(x, cond, labels)
    torch.cuda.synchronize()
    assert torch.equal(tr.grads, tr_ref.grads), it          # one rank: the sum is the gradient itself, every piece in place
tr.close(); tr_ref.close()
dist.barrier()
torch.cuda.synchronize()
print(f"RCCL OK backend={dist.get_backend()} world={dist.get_world_size()} broadcast {t_b*1e3:.1f} ms all_reduce {t_r*1e3:.1f} ms (first calls, incl. communicator set-up)")
dist.destroy_process_group()
