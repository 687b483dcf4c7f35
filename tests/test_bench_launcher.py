"""CPU: `python bench.py --gpus N` starts N ranks itself (BASELINE configs[3] must never silently run on one GPU).

The dry mode runs the launcher, the weight broadcast and the token all-gather on CPU tensors over gloo;
the GPU work of the ranks is what tests/test_*_gpu.py and the driver's SCALE run cover."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _run(extra_env=None, gpus=2):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(gpus), "--dry-run", "--batch", "3"],
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
