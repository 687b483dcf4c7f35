#!/bin/bash
# round 3: LDS-DMA GEMM — parity against the register-staged loop, then the training step / encoder with each setting
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gemm_dma_gpu.py -q -m gpu -s -p no:cacheprovider > gpurun_out/r3_dma_test.log 2>&1 || { tail -30 gpurun_out/r3_dma_test.log; exit 1; }
tail -3 gpurun_out/r3_dma_test.log
for S in "M2M_GEMM_DMA=0 M2M_TRAIN_KT_EPI=0" "M2M_GEMM_DMA=0" "M2M_GEMM_DMA=1" "M2M_GEMM_DMA_NS1=3" "M2M_GEMM_DMA_MINK=1024"; do
  echo "== $S"
  env $S timeout -k 10 300 python tools/train_bench.py bf16 dropout 2>&1 | grep -v amdgpu.ids
  env $S timeout -k 10 300 python tools/enc_bench.py 2>&1 | grep bf16
done
