#!/bin/bash
# Round 6: what dec_ff_multi_kernel's time consists of at 2 x 64 clips - rocprofv3 kernel averages of ablation builds beside the product build:
#   for m in 1 2 4 7; do M2M_BUILD_TAG=ffa$m M2M_BUILD_EXTRA="-DM2M_FF_ABL=$m" python -m music2midi_amd.csrc.build; done
# (1 = no atomics, 2 = no row reads, 4 = later slices re-use the first slice's weights)
OUT=gpurun_out/r6s; mkdir -p $OUT; export TMPDIR=/tmp
L=$PWD/music2midi_amd/lib
for t in product ffa1 ffa2 ffa4 ffa7; do
  lib=$L/libmusic2midi_amd_$t.so; [ $t = product ] && lib=$L/libmusic2midi_amd.so
  M2M_LIBRARY=$lib rocprofv3 --kernel-trace -d $OUT/$t -o np -- python3 tools/native_prof.py 128 512 > $OUT/$t.log 2>&1
  echo "== $t"; python3 tools/kernel_stats.py $OUT/$t 4; rm -rf $OUT/$t
done
