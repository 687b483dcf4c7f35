"""Multi-GPU plumbing: one process per GPU, clips sharded by contiguous blocks.

The path shards naturally — every clip has its own encoder pass, KV cache and
EOS state (ref: music2midi/model.py:115-135 already treats chunks
independently) — so there is no collective inside the decode loop.  Two
collectives exist, both outside it:

* a one-time broadcast of the weights from rank 0 as ONE flat buffer (RCCL
  broadcast over xGMI; a single large message instead of ~150 small ones), and
* one all-gather of the decoded token matrix per batch (<= 2 MiB in total), so
  every rank returns all rows in original clip order.

``backend="nccl"`` is RCCL on ROCm; ``gloo`` runs the same code on CPU tensors
(tests/test_distributed.py, world_size 2).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process if unset)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend: Optional[str] = None) -> Tuple[int, int, int]:
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        backend = os.environ.get("M2M_DIST_BACKEND", backend)
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            if torch.cuda.is_available():
                torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dist.init_process_group(backend)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
    return rank, local_rank, world


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition: rank r owns [lo, hi); sizes differ by at most one."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _gemm_weight(name: str, t: torch.Tensor) -> bool:
    """The tensors the bf16 mode's kernels read as bf16 GEMM operands (every 2-D weight except the token embedding, the
    relative-position-bias tables and the conditioning tables, which stay fp32 on the device): the set oracle/t5.py rounds."""
    return t.dim() == 2 and not any(k in name for k in ("relative_attention_bias", "shared", "embed_tokens", "conditioning", "mel_scale", "spectrogram"))


def broadcast_module_state(module: torch.nn.Module, src: int = 0, gemm_dtype: Optional[torch.dtype] = None) -> int:
    """Broadcast every parameter and buffer of ``module`` from ``src``.

    Default: ONE flat fp32 buffer (121.6 MB for the reference's model: the master weights travel, every rank repacks them
    identically, either precision mode can follow).  ``gemm_dtype=torch.bfloat16`` — for a run that decodes in the bf16 mode —
    sends the GEMM weights (``_gemm_weight``: 30.2 M of the 30.4 M parameters) as bfloat16 and the rest in fp32: two buffers,
    61.4 MB, the size SURVEY C4 gives the broadcast.  Receiving ranks hold the bf16 values upcast to fp32; repacking rounds
    to bf16 again, which is the identity on them, so every rank's device weights are bit-identical to rank 0's — and to a
    single-GPU run (rank 0 keeps its fp32 masters).  Returns the number of bytes sent.  No-op without a process group.
    """
    named = list(module.named_parameters(remove_duplicate=False)) + list(module.named_buffers(remove_duplicate=False))
    seen, uniq = set(), []
    for name, t in named:   # shared tensors (embed_tokens aliases) travel once
        if t.data_ptr() not in seen:
            seen.add(t.data_ptr())
            uniq.append((name, t))
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    dev = uniq[0][1].device
    groups = [(torch.float32, uniq)]
    if gemm_dtype is not None and gemm_dtype != torch.float32:
        groups = [(gemm_dtype, [(n, t) for n, t in uniq if _gemm_weight(n, t)]), (torch.float32, [(n, t) for n, t in uniq if not _gemm_weight(n, t)])]
    sent = 0
    is_src = dist.get_rank() == src
    for dt, items in groups:
        if not items:
            continue
        flat = torch.cat([t.detach().reshape(-1).to(dt) for _, t in items]).to(dev)
        dist.broadcast(flat, src=src)
        sent += flat.numel() * flat.element_size()
        if is_src and dt != torch.float32:
            continue                      # the source keeps its fp32 masters (its repack rounds them the same way)
        off = 0
        with torch.no_grad():
            for _, t in items:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t).to(t.dtype))
                off += n
    return sent


def module_checksum(module: torch.nn.Module, gemm_dtype: Optional[torch.dtype] = None) -> int:
    """64-bit position-weighted checksum of a module's parameters and buffers AS A RUN IN ``gemm_dtype`` WILL REPACK THEM (GEMM
    weights rounded to that type, everything else fp32): sum of word_i * (2 i + 1) mod 2^64 over the 16-bit / 32-bit words, tensor
    after tensor in ``named_parameters`` + ``named_buffers`` order.  Integer arithmetic on the tensors' own device (plumbing, no
    kernel of the path).  Equal on two ranks iff their replicas repack to the same bytes — the host-side twin of
    ``T5Transformer.device_weights_checksum`` (which sums the library's packed blob itself), usable without a GPU (gloo tests)."""
    named = list(module.named_parameters(remove_duplicate=False)) + list(module.named_buffers(remove_duplicate=False))
    seen, total, base = set(), 0, 0
    mask = (1 << 64) - 1
    for name, t in named:
        if t.data_ptr() in seen:
            continue
        seen.add(t.data_ptr())
        x = t.detach().reshape(-1)
        if gemm_dtype is not None and gemm_dtype != torch.float32 and _gemm_weight(name, t):
            words = x.to(gemm_dtype).contiguous().view(torch.int16).to(torch.int64) & 0xFFFF
        elif x.dtype == torch.float32:
            words = x.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        else:
            words = x.to(torch.float64).contiguous().view(torch.int64)
        n = words.numel()
        if n == 0:
            continue
        idx = torch.arange(base, base + n, dtype=torch.int64, device=words.device)
        # int64 multiply and sum wrap modulo 2^64, which is the arithmetic wanted
        total = (total + int((words * (2 * idx + 1)).sum().item())) & mask
        base += n
    return total


def tensor_checksum(t: torch.Tensor) -> int:
    """The same position-weighted 64-bit checksum over ONE tensor's 32-bit words (the trainer's flat fp32 parameter buffer): after K
    data-parallel steps every rank must hold the same bytes — its clips differ, so only a correct gradient average keeps them so."""
    x = t.detach().reshape(-1)
    words = (x.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF) if x.dtype == torch.float32 else x.to(torch.float64).contiguous().view(torch.int64)
    idx = torch.arange(words.numel(), dtype=torch.int64, device=words.device)
    return int((words * (2 * idx + 1)).sum().item()) & ((1 << 64) - 1)


def verify_replicas(checksums: Dict[str, int], device) -> Dict[str, object]:
    """All-reduce MIN and MAX of every named 64-bit checksum over the ranks and raise when a pair differs: a rank whose weights
    are not rank 0's (a broadcast that did not reach it, a different repack) must stop the run, not decode different tokens.
    Returns {"ranks_seen": world, "checksums": {name: hex}, "identical": True} — what bench.py prints at N > 1 so that the first
    multi-GPU run proves RCCL saw N ranks with identical replicas."""
    names = sorted(checksums)
    vals = [int(checksums[k]) & ((1 << 64) - 1) for k in names]
    signed = [v - (1 << 64) if v >= (1 << 63) else v for v in vals]          # two's complement: int64 tensors order consistently
    world = 1
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        world = dist.get_world_size()
        dev = torch.device("cpu") if dist.get_backend() == "gloo" else device
        lo = torch.tensor(signed + [1], dtype=torch.int64, device=dev)
        hi = lo.clone()
        cnt = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        world = int(cnt.item())
        bad = [k for k, a, b in zip(names, lo[:-1].tolist(), hi[:-1].tolist()) if a != b]
        if bad:
            raise RuntimeError(f"weight replicas differ between ranks after the broadcast: {bad} (this rank: "
                               + ", ".join(f"{k}={checksums[k] & ((1 << 64) - 1):#018x}" for k in bad) + ")")
        if world != dist.get_world_size():
            raise RuntimeError(f"collective saw {world} ranks, the process group says {dist.get_world_size()}")
    return {"ranks_seen": world, "checksums": {k: f"{v:#018x}" for k, v in zip(names, vals)}, "identical": True}


def all_gather_floats(value: float, device) -> List[float]:
    """One float per rank, in rank order (per-rank step times of a scaling run)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    dev = torch.device("cpu") if dist.get_backend() == "gloo" else device
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    parts = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [float(p.item()) for p in parts]


def all_gather_tokens(tokens: torch.Tensor, max_length: int, pad_id: int = 0) -> torch.Tensor:
    """Local [B_local, L_local] ids -> global [sum B_local, L_global] in rank (= clip) order.

    Rows are right-padded with ``pad_id`` to ``max_length`` for the collective, then trimmed to
    the longest valid length over all ranks — the shape one process decoding the whole batch
    would have returned (generation stops when EVERY row has finished).
    """
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tokens
    world = dist.get_world_size()
    B, L = tokens.shape
    send = torch.full((B, max_length + 1), pad_id, dtype=torch.long, device=tokens.device)
    send[:, :L] = tokens
    send[:, max_length] = L              # carry the local valid length in the last column (an empty shard carries nothing)
    counts = [torch.zeros(1, dtype=torch.long, device=tokens.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([B], dtype=torch.long, device=tokens.device))
    sizes = [int(c.item()) for c in counts]
    if len(set(sizes)) == 1:
        out = torch.empty((world * B, max_length + 1), dtype=torch.long, device=tokens.device)
        dist.all_gather_into_tensor(out, send)
    else:                                # ragged last shard: pad to the largest block
        bmax = max(sizes)
        padded = torch.full((bmax, max_length + 1), pad_id, dtype=torch.long, device=tokens.device)
        padded[:B] = send
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded)
        out = torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0)
    L_global = int(out[:, max_length].max().item()) if out.shape[0] else 1
    return out[:, :L_global].contiguous()


def generate_sharded(generate_fn, inputs, max_length: int, pad_id: int = 0):
    """Decode a batch of clips across all ranks: rank r runs ``generate_fn`` on its contiguous block of clips
    (``shard_range``) and the token matrices are all-gathered back into clip order — what one process decoding the
    whole batch would have returned.  ``inputs`` is a ``ModelInputs`` whose tensors have the FULL batch on every rank
    (clips are cheap to replicate: 0.9 MB each); without a process group this is ``generate_fn(inputs, max_length=...)``.
    Used by ``Music2MIDI.sample_tokens`` and by bench.py, so callers and the benchmark share one sharding path."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return generate_fn(inputs, max_length=max_length)
    rank, world = dist.get_rank(), dist.get_world_size()
    n = inputs.input_waveform.shape[0]
    lo, hi = shard_range(n, rank, world)
    if hi > lo:
        local = type(inputs)(input_waveform=inputs.input_waveform[lo:hi],
                             notes_batch=inputs.notes_batch[lo:hi] if inputs.notes_batch is not None else None,
                             cond_index=inputs.cond_index[lo:hi] if inputs.cond_index is not None else None)
        toks = generate_fn(local, max_length=max_length)
    else:       # more ranks than clips: this rank contributes nothing
        toks = torch.zeros((0, 1), dtype=torch.long, device=inputs.input_waveform.device)
    return all_gather_tokens(toks, max_length, pad_id)


def all_reduce_gradients(flat_grads: torch.Tensor) -> int:
    """Data-parallel gradient averaging (what Lightning's DDP does for ref train.py:40-41): ONE all-reduce of the
    flat fp32 gradient buffer (121.6 MB for the reference model; a ring all-reduce is a reduce-scatter + all-gather
    over xGMI), then 1/world.  Every rank then applies the identical Adafactor step to its replica.  Returns bytes reduced."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
    flat_grads.div_(dist.get_world_size())
    return flat_grads.numel() * 4


def split_ranges(n: int, early: Sequence[Tuple[int, int]]) -> Tuple[List[Tuple[int, int]], List[Tuple[int, int]]]:
    """(early, late) as (offset, count) lists that tile [0, n): `late` is the complement of the early ranges."""
    early = sorted((int(o), int(c)) for o, c in early if c > 0)
    late, pos = [], 0
    for o, c in early:
        if o < pos or o + c > n:
            raise ValueError(f"early ranges {early} overlap or leave [0, {n})")
        if o > pos:
            late.append((pos, o - pos))
        pos = o + c
    if pos < n:
        late.append((pos, n - pos))
    return early, late


def all_reduce_gradients_overlapped(flat_grads: torch.Tensor, early_ranges: Sequence[Tuple[int, int]], sync_stream=None) -> int:
    """The same averaging as ``all_reduce_gradients`` in pieces, for a trainer with a sync stream
    (``NativeTrainer.set_sync_stream``): the early ranges — final when the trainer releases `sync_stream`, half-way through the
    backward pass — are reduced on that stream while the encoder-side backward still runs; the rest follows behind the whole pass
    on the current stream, which then waits for the early pieces.  Call it right after ``forward_backward`` returns (that call
    only enqueues).  xGMI is point to point, so the pieces stay large: four collectives per step, the two early ones 2/3 of the
    bytes.  Returns bytes reduced."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    early, late = split_ranges(flat_grads.numel(), early_ranges)
    pending = []
    if sync_stream is not None and flat_grads.is_cuda:
        with torch.cuda.stream(sync_stream):
            for o, c in early:
                pending.append(dist.all_reduce(flat_grads[o:o + c], op=dist.ReduceOp.SUM, async_op=True))
    else:
        late = early + late
    for o, c in late:
        dist.all_reduce(flat_grads[o:o + c], op=dist.ReduceOp.SUM)
    for w in pending:
        w.wait()                       # the current stream continues behind the early pieces
    flat_grads.div_(world)
    return flat_grads.numel() * 4


def reduce_logged(metrics: Dict[str, object], device=None) -> Dict[str, float]:
    """What ``self.log(name, value, sync_dist=True)`` does in the reference (ref: music2midi/model.py:37,42,49,52): every
    logged scalar becomes its MEAN over the ranks (Lightning's default ``reduce_fx="mean"`` with ``sync_dist=True``).  All
    values travel in ONE all-reduce (a handful of doubles: SURVEY.md C2); values may be Python numbers or 0-dim tensors that
    are still on the device (this is where they are read — one host sync for the whole set).  ``batch_size`` is a count, not
    a metric: it is summed.  Without a process group the values are just converted."""
    names = sorted(metrics)
    if not names:
        return {}
    vals = [metrics[k] for k in names]
    on_dev = [v for v in vals if torch.is_tensor(v)]
    dev = device if device is not None else (on_dev[0].device if on_dev else torch.device("cpu"))
    packed = torch.stack([v.detach().to(dev, torch.float64).reshape(()) if torch.is_tensor(v) else torch.tensor(float(v), dtype=torch.float64, device=dev)
                          for v in vals])
    world = 1
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        world = dist.get_world_size()
        if dist.get_backend() == "nccl" and packed.device.type != "cuda":
            packed = packed.cuda()
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    out = packed.cpu().tolist()
    return {k: (v if k == "batch_size" else v / world) for k, v in zip(names, out)}


def is_rank_zero() -> bool:
    """Global rank 0 of the process group (True in a single process): the rank that writes checkpoints, as in Lightning."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank() == 0
    return env_world()[0] == 0 or env_world()[2] == 1


def barrier() -> None:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def broadcast_flag(flag: int, device, src: int = 0) -> int:
    """Rank ``src``'s integer flag on every rank (a collective: every rank calls it).  The exit of a rank-0-only action — unlike a
    bare barrier it tells the other ranks whether the action FAILED, so they raise with rank 0 instead of running into the next
    collective without it."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(flag)
    dev = torch.device("cpu") if dist.get_backend() == "gloo" else device
    t = torch.tensor([int(flag)], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=src)
    return int(t.item())


def all_reduce_max(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_reduce_sum(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
