"""CPU: `python bench.py --gpus N` starts N ranks itself (BASELINE configs[3] must never silently run on one GPU).

The dry mode runs the launcher, the weight broadcast and the token all-gather on CPU tensors over gloo;
the GPU work of the ranks is what tests/test_*_gpu.py and the driver's SCALE run cover."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _run(extra_env=None, gpus=2, extra_args=()):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(gpus), "--dry-run", "--batch", "3", *extra_args],
                          env=env, capture_output=True, text=True, timeout=300)


def test_bench_spawns_one_rank_per_gpu_and_rank0_prints_one_json_line():
    r = _run()
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["world"] == 2 and rec["backend"] == "gloo" and rec["dry_run"] is True
    assert rec["gather_ok"] and rec["weights_identical_on_all_ranks"]
    assert rec["config"]["global_batch"] == 6
    assert rec["config"]["weight_broadcast_bytes"] > 30_000_000 * 4       # every fp32 parameter travelled once
    # first-multi-GPU-run insurance: the line proves the collectives saw 2 ranks, carries the replicas' 64-bit
    # checksums (MIN == MAX over the ranks, or the run would have failed) and every rank's own step time
    assert rec["ranks_seen"] == 2 and rec["per_rank_ms_per_step"] == [1.0, 2.0]
    assert set(rec["replica_checksums"]) == {"masters_fp32", "as_bf16_repack"}
    assert all(v.startswith("0x") and len(v) == 18 for v in rec["replica_checksums"].values())
    assert rec["replica_checksums"]["masters_fp32"] != rec["replica_checksums"]["as_bf16_repack"]


def test_bench_replica_checksum_catches_a_rank_with_different_weights():
    """One weight of rank 1 changed after the broadcast (test hook): the MIN / MAX all-reduce of the checksums differs and EVERY
    rank stops with the mismatch named — diverged replicas never reach the timed region."""
    r = _run({"M2M_BENCH_CORRUPT_RANK": "1"})
    assert r.returncode != 0
    assert "weight replicas differ between ranks after the broadcast" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_launcher_fails_when_a_rank_dies():
    r = _run({"M2M_BENCH_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert "rank 1 exited" in r.stderr


def test_bench_honours_the_torchrun_environment():
    # the driver's way: RANK/LOCAL_RANK/WORLD_SIZE already set -> no second level of processes
    env = {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29517"}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--dry-run", "--batch", "2"],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_bench_watchdog_ends_a_hung_launch_and_names_the_live_ranks(tmp_path):
    """A rank stuck in communicator set-up is the likeliest failure of a first 8-GPU run: the launcher must not wait for the
    driver's timeout.  Rank 1 hangs (test hook); the parent gives up after M2M_BENCH_TIMEOUT, names it, shows its log, ends it."""
    import time
    t0 = time.monotonic()
    r = _run({"M2M_BENCH_HANG_RANK": "1", "M2M_BENCH_TIMEOUT": "20", "M2M_BENCH_LOG_DIR": str(tmp_path)})
    assert r.returncode == 124, (r.returncode, r.stderr[-1500:])
    assert time.monotonic() - t0 < 120
    assert "watchdog: ranks [0, 1] still running" in r.stderr             # rank 0 waits for rank 1 inside the collective: both are named
    assert "hanging on purpose" in r.stderr                              # the tail of the live rank's own log is shown
    assert (tmp_path / "bench_rank0.err").exists() and "hanging on purpose" in (tmp_path / "bench_rank1.err").read_text()


def test_bench_train_mode_dry_run_averages_gradients_and_metrics_over_the_ranks(tmp_path):
    """`--mode train --gpus 2 --dry-run`: BASELINE configs[4]'s data-parallel plumbing on gloo — the 121.6 MB flat gradient
    averaged in one piece and in the overlapped form's pieces, the sync_dist metric mean (ref model.py:37), pinned host threads."""
    r = _run({"M2M_BENCH_LOG_DIR": str(tmp_path)}, extra_args=("--mode", "train"))
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["mode"] == "train" and rec["world"] == 2 and rec["backend"] == "gloo" and rec["grad_average_ok"] is True
    assert rec["grad_allreduce_bytes"] > 30_000_000 * 4
    assert rec["logged"]["train/loss"] == 1.5 and rec["logged"]["train/score"] == 0.125 and rec["logged"]["batch_size"] == 32
    assert 1 <= rec["host_threads"] <= max(1, (os.cpu_count() or 2) // 2)
    # round 6: the training line carries at N > 1 what the generate line does, so the first 8-GPU run of configs[4]
    # proves by itself that the collectives saw N ranks and left identical replicas behind
    assert rec["ranks_seen"] == 2 and len(rec["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in rec["per_rank_ms_per_step"])
    assert rec["grad_allreduce_bytes_per_step"] == rec["grad_allreduce_bytes"]
    assert rec["replica_checksums"]["params_after_steps"].startswith("0x")


def test_bench_train_mode_catches_replicas_that_differ_after_the_steps(tmp_path):
    """One rank's parameters changed after the averaged step (test hook): the MIN / MAX all-reduce of the after-steps checksum differs,
    every rank stops, no line is printed."""
    r = _run({"M2M_BENCH_LOG_DIR": str(tmp_path), "M2M_BENCH_CORRUPT_RANK": "1"}, extra_args=("--mode", "train"))
    assert r.returncode != 0
    assert "replicas differ between ranks" in r.stderr and "params_after_steps" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_pmc_summary_without_the_launch_width_header_is_refused(tmp_path, capsys):
    sys.path.insert(0, str(ROOT))
    import bench
    line = ("m2m::dec_attn_kernel<m2m::bf16_t, false, 1>   launches    288  FETCH_SIZE/launch    14462.0 KiB (x2 corrected    29.62 MB)  "
            "WRITE_SIZE/launch     430.0 KiB ( 0.440 MB)\n")
    (tmp_path / "r3_x_pmc_traffic_summary.txt").write_text(line)
    assert bench.pmc_traffic_bytes("dec_attn_kernel<m2m::bf16_t, false,", 32, profiles_dir=tmp_path) is None
    assert "clips/launch" in capsys.readouterr().err
    (tmp_path / "r3_x_pmc_traffic_summary.txt").write_text("# decode kernels: clips/launch = 16\n" + line)
    got = bench.pmc_traffic_bytes("dec_attn_kernel<m2m::bf16_t, false,", 32, profiles_dir=tmp_path)
    assert abs(got - 2 * (29.62 + 0.44) * 1e6) < 1.0                       # two 16-clip launches make the 32-clip launch set
    # the committed summary of the newest round carries the header and gives ~1.06x the algorithmic 56.7 MB
    real = bench.pmc_traffic_bytes("dec_attn_kernel<m2m::bf16_t, false,", 32)
    assert real is None or 0.9 * 56.72e6 < real < 1.3 * 56.72e6, real


def test_train_profile_summary_reads_the_committed_rocprof_stats():
    """train_configs4.roofline takes launches per step and the non-product share of a step from the newest committed
    profiles/r*_train_dropout_kernel_stats.csv (rocprofv3 --kernel-trace --stats of tools/train_gap.py, 23 directly issued steps)."""
    sys.path.insert(0, str(ROOT))
    import bench
    rec = bench.train_profile_summary()
    assert rec and rec["profile"].startswith("r") and rec["profile"].endswith("train_dropout_kernel_stats.csv")
    assert 150 < rec["launches_per_step"] < 400 and 3000 < rec["kernel_us_per_step"] < 6000
    assert 0.15 * rec["kernel_us_per_step"] < rec["non_gemm_us"] < 0.5 * rec["kernel_us_per_step"]
    assert bench.train_profile_summary(profiles_dir=ROOT / "tests") == {}
