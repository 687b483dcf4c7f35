#!/bin/bash
# per-(kernel, grid) durations of the training step's projection GEMMs with and without the LDS-DMA loop
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/bygrid
for D in 0 1; do
  M2M_GEMM_DMA=$D M2M_GAP_DROPOUT=0.1 M2M_TRAIN_GRAPH=0 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bygrid/d$D -o t -- python3 tools/train_gap.py > gpurun_out/bygrid/run$D.log 2>&1
  grep WALL gpurun_out/bygrid/run$D.log
  F=$(find gpurun_out/bygrid/d$D -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_by_grid.py $F gemm_kernel > gpurun_out/bygrid/gemm_d$D.txt
  rm -rf gpurun_out/bygrid/d$D
done
paste -d'\n' gpurun_out/bygrid/gemm_d0.txt gpurun_out/bygrid/gemm_d1.txt | cut -c1-200
