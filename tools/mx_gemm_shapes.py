"""Kernel durations of the projection products at the training step's shapes (16 clips: M = 4176 encoder rows / 4096 decoder rows):
the pre-quantised MXFP8 product, the product that quantises A in its staging, each against the same shape through the step itself
(bf16: profiles/r2_*_train_kernel_stats.csv).  Run under `rocprofv3 --kernel-trace --stats`; the per-kernel averages are the result.
  python tools/mx_gemm_shapes.py [reps]"""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from music2midi_amd import native  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lib = native.load()
dev = torch.device("cuda:0")
st = native.stream_handle(dev)
shapes = [(4176, 1152, 384), (4176, 2048, 384), (4176, 384, 1024), (4176, 384, 384), (4176, 384, 2048), (4176, 384, 1152), (4096, 768, 384)]
for (M, N, K) in shapes:
    a = (torch.randn(M, K, device=dev)).to(torch.bfloat16).contiguous()
    b = torch.randn(N, K, device=dev).contiguous()
    c = torch.empty(M, N, device=dev)
    for fused in (0, 1):
        for _ in range(reps):
            native.check(lib.m2m_mx8_matmul_bf16a(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), M, N, K, 0, fused, C.c_void_p(c.data_ptr()), st), "mx8")
    torch.cuda.synchronize()
    print(M, N, K, "done", flush=True)
