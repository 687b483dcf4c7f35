"""Times the row-complete residual product (csrc/rowgemm_train.hip) alone: python tools/rowgemm_bench.py [M]   (default 4176 = 16 x 261)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from music2midi_amd import native

lib = native.load()
native.require_gpu()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4176
N = 384
key = torch.zeros(1, dtype=torch.int64, device="cuda")
for K in (512, 1152):
    A = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    x = torch.randn(M, N, device="cuda")
    w = torch.rand(N, device="cuda") + 0.5
    xo = torch.empty_like(x)
    h = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    st = native.stream_handle()
    for dbg, p in ((0, 0.0), (0, 0.1), (1, 0.0), (2, 0.0), (4, 0.0), (7, 0.0)):
        def f():
            native.check(lib.m2m_rowgemm_norm_bf16(A.data_ptr(), W.data_ptr(), x.data_ptr(), w.data_ptr(), M, N, K, 1e-6, float(p), key.data_ptr(), C.c_uint64(5),
                                                   xo.data_ptr(), h.data_ptr(), dbg, st), "rowgemm")
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 10
        print(f"M={M} N={N} K={K} dropout {p} dbg={dbg}: {us:6.1f} us  ({2.0 * M * N * K / us * 1e-6:6.1f} TFLOP/s)")
    if True:
        ref = x.double() + A.double() @ W.double().t()
        print("   max |x_out - ref| (dbg 0, p 0):", end=" ")
        native.check(lib.m2m_rowgemm_norm_bf16(A.data_ptr(), W.data_ptr(), x.data_ptr(), w.data_ptr(), M, N, K, 1e-6, 0.0, key.data_ptr(), C.c_uint64(5), xo.data_ptr(),
                                               h.data_ptr(), 0, st), "rowgemm")
        torch.cuda.synchronize()
        print(float((xo.double() - ref).abs().max()))
