#!/bin/bash
# Round 6: same-box A/B of the multi-clip decode attention against tagged builds of the library beside the product build, e.g.
#   M2M_BUILD_TAG=w4 M2M_BUILD_EXTRA="-DM2M_MC_WAVES=4" python -m music2midi_amd.csrc.build      (128 registers: one workgroup per CU)
#   M2M_BUILD_TAG=pf3 M2M_BUILD_EXTRA="-DM2M_MC_PF_SELF=3 -DM2M_MC_PF_CROSS=3" python -m music2midi_amd.csrc.build
# usage: tools/r6_mc_ab.sh OUT "tag ..." B T precision L spec...     (spec = clips,ffrows,grouprows; tag "product" = the in-tree library)
set -u
OUT=$1; TAGS=$2; shift 2
mkdir -p $OUT
L=$PWD/music2midi_amd/lib
for t in $TAGS; do
  lib=$L/libmusic2midi_amd_$t.so; [ $t = product ] && lib=$L/libmusic2midi_amd.so
  echo "== $t"
  M2M_LIBRARY=$lib python tools/native_mc_sweep.py "$@" 2>&1 | tee -a $OUT/ab_$t.log
done
