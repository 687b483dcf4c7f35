#!/bin/bash
# Round 6: what the large-chain decode kernels' time consists of - rocprofv3 kernel averages of ablation builds (tools/r6_mc_ab.sh for the build recipe;
# tags ffa1/2/3 = -DM2M_FF_ABL, mca1/2/3 = -DM2M_MC_ABL) at the reference-native shape, M2M_DA_CLIPS=2.
OUT=${1:-gpurun_out/r6c}; mkdir -p $OUT; export TMPDIR=/tmp
L=$PWD/music2midi_amd/lib
for t in product ffa1 ffa2 ffa3 mca1 mca2 mca3; do
  lib=$L/libmusic2midi_amd_$t.so; [ $t = product ] && lib=$L/libmusic2midi_amd.so
  M2M_LIBRARY=$lib M2M_DA_CLIPS=${CLIPS:-2} M2M_DEC_FF_ROWS=8 rocprofv3 --kernel-trace -d $OUT/$t -o np -- python3 tools/native_prof.py 128 1024 > $OUT/$t.log 2>&1
  echo "== $t"; python3 tools/kernel_stats.py $OUT/$t 6
done
