"""GPU: the LDS-DMA main loop of the projection GEMM (csrc/enc_kernels.hip gemm_kernel<..., NS>: operand tiles global -> LDS through
global_load_lds_dwordx4 into a ring of 64-deep stages, counted vmcnt + one raw barrier per step, bank swizzle on the source
address) against the register-staged loop it replaces.  Both accumulate every output element over k in the same order, so every
result downstream — encoder states, greedy ids, training loss and gradients — must be BIT-identical, at every ring depth.
The same runs cover the gated-GELU forward and backward written by the neighbouring products' own epilogues (EPI_GATED_TRAIN on
row-interleaved weights, EPI_GATED_BWD; M2M_TRAIN_GATE_EPI=0 restores the gated_fwd / gated_bwd kernel launches): the same
accumulations, the activation and its derivative computed from the same rounded values with the same functions — bit-identical
again (the 16-clip shape takes the fused forward; every shape the fused backward)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _run(env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "gemm_dma_check.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith(("tiny", "full"))]
    assert len(lines) >= 8, r.stdout
    return lines


def test_lds_dma_gemm_is_bit_identical_to_the_register_staged_gemm():
    ref = _run({"M2M_GEMM_DMA": "0", "M2M_TRAIN_GATE_EPI": "0"})       # register-staged loop everywhere, gated_fwd_kernel launches
    # default = the ring where it pays (small tiles, K >= 512); "all" = every bf16 product incl. the 128x128 tiles, at two ring depths
    for setting in ({}, {"M2M_GEMM_DMA": "all"}, {"M2M_GEMM_DMA": "all", "M2M_GEMM_DMA_NS1": "3", "M2M_GEMM_DMA_NS2": "2"}):
        got = _run(setting)
        for a, b in zip(ref, got):
            assert a == b, f"{setting or 'default ring depths'}:\n  register-staged: {a}\n  LDS-DMA:         {b}"
    print("\n".join(ref))
