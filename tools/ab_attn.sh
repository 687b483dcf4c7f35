#!/bin/bash
# same-box A/B of the encoder attention forms on the encoder bench (kernel averages under rocprofv3):
#   M2M_ATTN_WIDE=1 (default: 64-key steps, double-buffered tiles) against M2M_ATTN_WIDE=0 (the first kernel), twice each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in 1 0 1 0; do
  D=gpurun_out/ab_attn_wide$w
  rm -rf $D; M2M_ATTN_WIDE=$w rocprofv3 --kernel-trace --stats --output-format csv -d $D -o enc -- python3 tools/enc_bench.py > $D.log 2>&1
  echo "== M2M_ATTN_WIDE=$w: $(grep bf16 $D.log)"
  grep "attn_" $D/enc_kernel_stats.csv | cut -d, -f1,2,4,6,7 | cut -c1-140
done
