"""CPU: the N>1 path (shard -> broadcast -> all-gather -> original order) with gloo, world_size 2."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from music2midi_amd import distributed as D


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ragged, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, _, w = D.init_process_group("gloo")
    assert (r, w) == (rank, world)
    # 1. weight broadcast: rank 0's values win, aliases travel once
    torch.manual_seed(100 + rank)
    emb = torch.nn.Embedding(7, 4)
    mod = torch.nn.ModuleDict({"a": emb, "alias": emb, "l": torch.nn.Linear(4, 3)})
    mod.register_buffer("buf", torch.full((5,), float(rank)))
    nbytes = D.broadcast_module_state(mod, src=0)
    torch.manual_seed(100)
    ref_emb = torch.nn.Embedding(7, 4)
    ref_lin = torch.nn.Linear(4, 3)
    assert torch.equal(mod["a"].weight, ref_emb.weight) and torch.equal(mod["l"].weight, ref_lin.weight)
    assert torch.equal(mod.buf, torch.zeros(5)) and nbytes == (28 + 12 + 3 + 5) * 4
    # 2. clip sharding + token all-gather back into clip order
    n_clips, max_len = (7 if ragged else 8), 12
    lo, hi = D.shard_range(n_clips, rank, world)
    L_local = 5 + 3 * rank                      # ranks stop at different lengths
    toks = torch.zeros((hi - lo, L_local), dtype=torch.long)
    for i, clip in enumerate(range(lo, hi)):
        toks[i] = torch.arange(L_local) + 100 * clip
    out = D.all_gather_tokens(toks, max_len, pad_id=0)
    assert out.shape == (n_clips, 5 + 3 * (world - 1))
    for clip in range(n_clips):
        owner = next(r_ for r_ in range(world) if D.shard_range(n_clips, r_, world)[0] <= clip < D.shard_range(n_clips, r_, world)[1])
        L = 5 + 3 * owner
        assert torch.equal(out[clip, :L], torch.arange(L) + 100 * clip) and (out[clip, L:] == 0).all()
    assert D.all_reduce_max(float(rank), "cpu") == world - 1 and D.all_reduce_sum(1.0, "cpu") == world
    D.barrier()
    dist.destroy_process_group()
    q.put(rank)


@pytest.mark.parametrize("ragged", [False, True])
def test_gloo_world2_broadcast_and_token_gather(ragged):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ragged, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]


def test_single_process_paths_are_noops():
    t = torch.arange(6).reshape(2, 3)
    assert D.all_gather_tokens(t, 8) is t
    assert D.broadcast_module_state(torch.nn.Linear(2, 2)) == 0
    assert D.all_reduce_max(3.0, "cpu") == 3.0


def _grad_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    D.init_process_group("gloo")
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)          # rank r holds (r+1) * g
    nbytes = D.all_reduce_gradients(flat)
    assert nbytes == 4000 and torch.equal(flat, torch.arange(1000, dtype=torch.float32) * 1.5)   # mean over the 2 ranks
    dist.destroy_process_group()
    q.put(rank)


def test_gloo_world2_gradient_all_reduce_averages_the_flat_buffer():
    """Data-parallel training (BASELINE configs[4]): one collective per step on the flat gradient buffer."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]
    assert D.all_reduce_gradients(torch.ones(4)) == 0          # single process: no-op


def _grad_pieces_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    D.init_process_group("gloo")
    ref = torch.arange(1000, dtype=torch.float32)
    flat = ref * (rank + 1)
    nbytes = D.all_reduce_gradients_overlapped(flat, [(0, 130), (700, 250)])          # early: front + a block near the end
    assert nbytes == 4000 and torch.equal(flat, ref * 1.5)
    dist.destroy_process_group()
    q.put(rank)


def test_gloo_world2_gradient_all_reduce_in_pieces():
    """The overlapped form (early ranges on the trainer's sync stream, the rest behind the pass) averages every element once."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_pieces_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]


def test_split_ranges_tile_the_buffer():
    early, late = D.split_ranges(100, [(60, 30), (0, 10)])
    assert early == [(0, 10), (60, 30)] and late == [(10, 50), (90, 10)]
    assert D.split_ranges(10, [(0, 10)]) == ([(0, 10)], [])
    import pytest
    with pytest.raises(ValueError):
        D.split_ranges(10, [(0, 6), (5, 3)])
    with pytest.raises(ValueError):
        D.split_ranges(10, [(8, 3)])


def _shard_worker(rank, world, port, n_clips, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    D.init_process_group("gloo")
    from music2midi_amd.input import ModelInputs
    wav = torch.arange(n_clips, dtype=torch.float32)[:, None].repeat(1, 8)       # clip c carries the value c
    cond = torch.stack([torch.arange(n_clips) % 6, torch.arange(n_clips) % 3], dim=1)
    calls = []

    def fake_generate(inp, max_length):                                           # ids encode (clip, cond) so order/pairing is visible
        calls.append(inp.input_waveform.shape[0])
        L = 3 + int(inp.input_waveform[0, 0].item()) % 4                           # shards stop at different lengths
        out = torch.zeros((inp.input_waveform.shape[0], L), dtype=torch.long)
        out[:, 0] = 1
        out[:, 1] = inp.input_waveform[:, 0].long() + 10
        out[:, 2] = inp.cond_index[:, 0] * 10 + inp.cond_index[:, 1] + 100
        return out

    ids = D.generate_sharded(fake_generate, ModelInputs(input_waveform=wav, cond_index=cond), max_length=16)
    lo, hi = D.shard_range(n_clips, rank, world)
    assert calls == ([hi - lo] if hi > lo else [])
    assert ids.shape[0] == n_clips and torch.equal(ids[:, 1], torch.arange(n_clips) + 10)
    assert torch.equal(ids[:, 2], cond[:, 0] * 10 + cond[:, 1] + 100)
    dist.destroy_process_group()
    q.put(rank)


@pytest.mark.parametrize("n_clips", [5, 1])
def test_gloo_world2_generate_sharded_returns_clip_order(n_clips):
    """Music2MIDI.sample_tokens' multi-GPU path: clips sharded over the ranks, ids back in clip order, also when a rank has no clip."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, n_clips, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]


def _logged_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    D.init_process_group("gloo")
    out = D.reduce_logged({"train/loss": torch.tensor(2.0 + rank), "train/score": 0.5 * rank, "batch_size": 16})
    assert out == {"train/loss": 2.5, "train/score": 0.25, "batch_size": 32}, out
    assert D.reduce_logged({}) == {}
    dist.destroy_process_group()
    q.put(rank)


def test_gloo_world2_logged_metrics_are_averaged_like_sync_dist():
    """ref model.py:37,42,49,52: self.log(..., sync_dist=True) = the mean over the ranks, in one packed all-reduce."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_logged_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]
    assert D.reduce_logged({"val/loss": torch.tensor(3.0), "batch_size": 4}) == {"val/loss": 3.0, "batch_size": 4}      # single process


def _ckpt_worker(rank, world, port, path, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import copy
    from pathlib import Path
    from music2midi_amd.checkpoint import read_checkpoint
    from music2midi_amd.config import DEFAULT_CONFIG
    from music2midi_amd.model import Music2MIDI
    D.init_process_group("gloo")
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    cfg["model"]["t5"].update(d_model=128, d_ff=256, num_layers=1, num_decoder_layers=1, num_heads=2)
    torch.manual_seed(7 + rank)                      # the ranks hold DIFFERENT weights: whose file it is shows in the content
    m = Music2MIDI(cfg)
    m.global_step = 11 + rank
    assert D.is_rank_zero() == (rank == 0)
    for _ in range(3):                               # rewritten in place, as save_every_n_steps does
        m.save_checkpoint(path)
        ck = read_checkpoint(path)                   # every rank, right after the call: the barrier makes the finished file visible
        assert int(ck["global_step"]) == 11
        w = ck["state_dict"]["model.transformer.lm_head.weight"]
        torch.manual_seed(7)
        assert torch.equal(torch.as_tensor(w), Music2MIDI(cfg).model.transformer.lm_head.weight.detach())
    D.barrier()
    assert [p.name for p in Path(path).parent.iterdir()] == [Path(path).name]        # no temporary file left behind
    # rank 0 failing to write (here: the directory does not exist) must raise on EVERY rank — with a bare barrier the
    # others returned normally and hung in the next gradient all-reduce
    bad = str(Path(path).parent / "no_such_dir" / "x.ckpt")
    try:
        m.save_checkpoint(bad)
        raise AssertionError("save_checkpoint into a missing directory did not raise")
    except AssertionError:
        raise
    except Exception as e:
        assert (rank == 0) or "rank 0 failed" in str(e), e
    # the guarded calling pattern (`if rank == 0: save`) stays possible: collective=False makes no collective call
    if rank == 0:
        m.save_checkpoint(str(Path(path).parent / "guarded.ckpt"), collective=False)
    D.barrier()
    assert (Path(path).parent / "guarded.ckpt").exists()
    dist.destroy_process_group()
    q.put(rank)


def test_gloo_world2_checkpoint_is_written_by_rank_zero_only(tmp_path):
    """fit_batches(save_path=) called save_checkpoint on every data-parallel rank onto one path.  Now rank 0
    writes (temporary file + rename), every rank leaves through a broadcast of rank 0's outcome (round 5: a failed write raises on
    every rank); Lightning writes on global rank 0 only (ref train.py:40-41)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    path = str(tmp_path / "run.ckpt")
    procs = [ctx.Process(target=_ckpt_worker, args=(r, 2, port, path, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]


def _bf16_bcast_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import copy
    from music2midi_amd.config import DEFAULT_CONFIG
    from music2midi_amd.transformer import T5Transformer
    D.init_process_group("gloo")
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    cfg["model"]["t5"].update(d_model=128, d_ff=256, num_layers=1, num_decoder_layers=1, num_heads=2)
    torch.manual_seed(50 + rank)
    m = T5Transformer(cfg, precision="bf16")
    torch.manual_seed(50)
    ref = T5Transformer(cfg, precision="bf16")                  # rank 0's weights, rebuilt locally
    n_params = sum(p.numel() for p in m.parameters())
    sent = D.broadcast_module_state(m, src=0, gemm_dtype=torch.bfloat16)
    got, want = dict(m.named_parameters()), dict(ref.named_parameters())
    n_gemm = 0
    for k, w in want.items():
        gemm = D._gemm_weight(k, w)
        n_gemm += w.numel() if gemm else 0
        if rank == 0:
            assert torch.equal(got[k], w), k                                           # the source keeps its fp32 masters
        elif gemm:
            assert torch.equal(got[k], w.detach().bfloat16().float()), k               # bf16 on the wire, upcast on arrival
            assert torch.equal(got[k].bfloat16(), w.detach().bfloat16()), k            # -> the repacked device weights are bit-identical
        else:
            assert torch.equal(got[k], w), k                                           # embeddings / norms / tables: exact fp32
    assert any(D._gemm_weight(k, w) for k, w in want.items()) and not D._gemm_weight("transformer.shared.weight", want["transformer.shared.weight"])
    n_buf = sum(b.numel() for b in m.buffers())
    assert sent == 2 * n_gemm + 4 * (n_params - n_gemm + n_buf), (sent, n_gemm, n_params, n_buf)
    dist.destroy_process_group()
    q.put(rank)


def test_gloo_world2_bf16_weight_broadcast_repacks_bit_identically():
    """SURVEY C4 sizes the inference weight broadcast at bf16 (60.8 MB); round 3 sent the fp32 masters (121.6 MB).  With
    gemm_dtype=bfloat16 the GEMM weights travel as bf16 and everything the device keeps in fp32 travels as fp32: the receiving
    ranks' repacked weights equal rank 0's bit for bit (rounding to bf16 is idempotent)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bf16_bcast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]
