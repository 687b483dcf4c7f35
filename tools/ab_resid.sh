#!/bin/bash
# same-box A/B of two builds of the library on the encoder bench (per-kernel averages under rocprofv3)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for tag in "" rpscalar "" rpscalar; do
  L=$PWD/music2midi_amd/lib/libmusic2midi_amd${tag:+_$tag}.so
  D=gpurun_out/ab_${tag:-product}
  rm -rf $D; M2M_LIBRARY=$L rocprofv3 --kernel-trace --stats --output-format csv -d $D -o enc -- python3 tools/enc_bench.py > $D.log 2>&1
  echo "== ${tag:-product}: $(grep bf16 $D.log)"
  grep "resid_panel\|norm_gemm" $D/enc_kernel_stats.csv | cut -d, -f1,2,4,6,7 | cut -c1-120
done
