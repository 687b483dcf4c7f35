#!/bin/bash
# parity of the GEMM / epilogue variants, then the 16-clip step time per setting given as arguments (default: the product defaults)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gemm_dma_gpu.py -q -m gpu -s -p no:cacheprovider > gpurun_out/r3_dma_test.log 2>&1 || { tail -30 gpurun_out/r3_dma_test.log; exit 1; }
tail -2 gpurun_out/r3_dma_test.log
[ $# -eq 0 ] && set -- "M2M_X=0"
for S in "$@"; do
  echo "== $S"
  env $S timeout -k 10 300 python tools/train_bench.py bf16 dropout 2>&1 | grep -v amdgpu.ids
done
