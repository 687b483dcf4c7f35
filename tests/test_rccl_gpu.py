"""The RCCL collectives of the multi-GPU paths on a one-rank nccl group (tools/rccl_smoke.py): weight broadcast, token
all-gather, gradient all-reduce, scalar reductions, barrier — so that the environment the 8-GPU run depends on is checked
on the 1-GPU box.  (The sharding logic itself is covered on CPU with gloo, world_size 2: tests/test_distributed.py.)"""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_rccl_collectives_on_one_rank():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "rccl_smoke.py")], capture_output=True, text=True, timeout=600, env=env)
    print(r.stdout[-500:])
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    assert "RCCL OK" in r.stdout
