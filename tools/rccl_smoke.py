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
dist.barrier()
torch.cuda.synchronize()
print(f"RCCL OK backend={dist.get_backend()} world={dist.get_world_size()} broadcast {t_b*1e3:.1f} ms all_reduce {t_r*1e3:.1f} ms (first calls, incl. communicator set-up)")
dist.destroy_process_group()
