"""Differential fuzz of the greedy decode loop (tools/fuzz_decode.py): random batch / encoder length / max_length / EOS
behaviour / chain split / graph length against the CPU oracle in fp32 — ids identical, near-ties excused only by the
oracle's own top-2 margin."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.parametrize("seed", [3, 19])
def test_decode_fuzz_against_oracle(seed):
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "fuzz_decode.py"), "100", str(seed)], capture_output=True, text=True, timeout=1500)
    tail = "\n".join(r.stdout.splitlines()[-6:])
    print(tail)
    assert r.returncode == 0, tail + r.stderr[-2000:]
    assert "FUZZ OK" in r.stdout


@pytest.mark.parametrize("seed", [1, 7])
def test_training_step_fuzz_against_autograd(seed):
    """Differential fuzz of the training step (tools/fuzz_train.py): random batch / encoder / label lengths around the tile edges
    of the fused attention kernels and past their reach, ignored labels, dropout on and off — every gradient tensor within 1e-4 of
    autograd over the oracle (fp32 mode), graph replay identical to the direct issue."""
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "fuzz_train.py"), "40", str(seed)], capture_output=True, text=True, timeout=1500)
    tail = "\n".join(r.stdout.splitlines()[-6:])
    print(tail)
    assert r.returncode == 0, tail + r.stderr[-2000:]
    assert "FUZZ OK" in r.stdout


@pytest.mark.parametrize("prec,seed", [("bf16", 5), ("fp8", 2)])
def test_training_step_fuzz_reduced_precision(prec, seed):
    """The same fuzz in the bf16 and MXFP8 modes: every loss and gradient finite, loss within 1 % / 5 %, every gradient tensor
    within a relative L2 of 0.12 (bf16, against fp32 autograd; two-label cases reach 0.106) / 0.6 (fp8, against the oracle that quantises its projections
    the same way) — the bounds catch NaNs, dropped terms and stale buffers at odd shapes, not rounding."""
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "fuzz_train.py"), "40", str(seed), prec], capture_output=True, text=True, timeout=1500)
    tail = "\n".join(r.stdout.splitlines()[-6:])
    print(tail)
    assert r.returncode == 0, tail + r.stderr[-2000:]
    assert "FUZZ OK" in r.stdout
