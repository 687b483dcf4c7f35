"""Oracle: OCP Microscaling FP8 (MXFP8) quantisation and products (test infrastructure, see oracle/__init__.py).

Restates the OCP Microscaling Formats v1.0 rule the device's fp8 training mode uses (csrc/mx8.hip): along the reduction
dimension every 32 consecutive elements share a power-of-two scale 2^(floor(log2(amax)) - emax_elem) (emax_elem = 8 for
e4m3fn, 15 for e5m2; an all-zero block gets 2^-127), the elements are cast to FP8 (round to nearest even, clamped to the
format's largest finite value).  torch's own float8_e4m3fn / float8_e5m2 casts do the element rounding, so a product of
dequantised operands in fp32 is what gfx950's scaled MFMA computes up to the order of the fp32 accumulation.
"""
from __future__ import annotations

import torch

_FMT = {"e4m3": (torch.float8_e4m3fn, 8, 448.0), "e5m2": (torch.float8_e5m2, 15, 57344.0)}


def mx_quant_dequant(x: torch.Tensor, fmt: str = "e4m3") -> torch.Tensor:
    """Quantise along the LAST dimension in blocks of 32 and dequantise again (fp32 in, fp32 out)."""
    dtype, emax, lim = _FMT[fmt]
    K = x.shape[-1]
    Kp = (K + 31) // 32 * 32
    xp = torch.nn.functional.pad(x.float(), (0, Kp - K))
    blocks = xp.reshape(*xp.shape[:-1], Kp // 32, 32)
    amax = blocks.abs().amax(dim=-1, keepdim=True)
    _, e = torch.frexp(amax)                                   # amax = m * 2^e, m in [0.5, 1)
    se = torch.where(amax > 0, (e - 1 - emax).float(), torch.full_like(amax, -127.0)).clamp(-127, 127)
    scale = torch.exp2(se)
    q = (blocks / scale).clamp(-lim, lim).to(dtype).float()
    return (q * scale).reshape(xp.shape)[..., :K]


def mx_matmul(a: torch.Tensor, b: torch.Tensor, a_fmt: str = "e4m3") -> torch.Tensor:
    """C[M,N] = A[M,K] . B[N,K]^T with both operands MX-quantised along K (B always e4m3)."""
    return mx_quant_dequant(a, a_fmt) @ mx_quant_dequant(b, "e4m3").T
