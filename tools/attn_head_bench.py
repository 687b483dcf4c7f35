"""Times the whole-head attention kernels of the training step (csrc/attn_train.hip) on their own, through the C ABI test entries:
    python tools/attn_head_bench.py [B ...]          (default 16; H = 8, the three shapes of a configs[4] step)
Prints us per launch (HIP events around `reps` launches on torch's current stream) and the matrix-core rate that corresponds to."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch

from music2midi_amd import native

lib = native.load()
native.require_gpu()


def run(B, H, Sq, Sk, causal, bias, p, reps=50):
    dev = "cuda"
    q = (torch.randn(B, Sq, H * 64, device=dev) * 0.6).bfloat16()
    k = (torch.randn(B, Sk, H * 64, device=dev) * 0.6).bfloat16()
    v = (torch.randn(B, Sk, H * 64, device=dev) * 0.6).bfloat16()
    do = (torch.randn(B, Sq, H * 64, device=dev) * 0.05).bfloat16()
    bt = (torch.randn(H, Sq + Sk - 1, device=dev) * 1.5).float() if bias else None
    out = torch.empty_like(q)
    lse = torch.empty(B * H, Sq, dtype=torch.float32, device=dev)
    bits = torch.zeros(B * H, (Sk + 31) // 32, (Sq + 31) // 32 * 32, dtype=torch.int32, device=dev)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    diag = torch.empty(B * H, (Sq + 31) // 32, Sk + 31, dtype=torch.float32, device=dev) if bias else None
    bp = bt.data_ptr() if bias else None
    st = native.stream_handle()

    def fwd():
        native.check(lib.m2m_attn_head_fwd_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), bp, B, H, Sq, Sk, int(causal), float(p), C.c_uint64(7), C.c_uint64(11),
                                                out.data_ptr(), lse.data_ptr(), bits.data_ptr(), st), "fwd")

    def bwd():
        native.check(lib.m2m_attn_head_bwd_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(), do.data_ptr(), bp, B, H, Sq, Sk,
                                                int(causal), float(p), C.c_uint64(7), C.c_uint64(11), bits.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                                                diag.data_ptr() if bias else None, st), "bwd")
    res = []
    for f in (fwd, bwd):
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / reps)
    pairs = B * H * Sq * Sk * (0.5 if causal else 1.0)
    fl_f, fl_b = 4 * 64 * pairs, 14 * 64 * pairs            # forward: 2 products; backward as built: 7 (S and dP twice)
    print(f"B={B:3d} H={H} Sq={Sq} Sk={Sk} causal={int(causal)} bias={int(bias)} p={p}: forward {res[0]:6.1f} us ({fl_f / res[0] * 1e-6:5.1f} TFLOP/s), "
          f"backward {res[1]:6.1f} us ({fl_b / res[1] * 1e-6:5.1f} TFLOP/s)")


if __name__ == "__main__":
    Bs = [int(x) for x in sys.argv[1:]] or [16]
    for B in Bs:
        run(B, 8, 261, 261, False, True, 0.1)
        run(B, 8, 256, 256, True, True, 0.1)
        run(B, 8, 256, 261, False, False, 0.1)
