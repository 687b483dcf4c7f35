#!/bin/bash
# Compile-time ablation of attn_wide_kernel's 64-key step: builds lib/libmusic2midi_amd_awcut<N>.so for each mask N (only
# enc_kernels.hip is recompiled, with -DM2M_AW_CUT=N; the other objects come from the product build) — run HERE (no GPU), then
# `gpurun -- tools/aw_variants.sh run` times every variant at B = 4 (one workgroup per CU: the lone-wave chain) and B = 32.
#   bits: 1 exponentials  2 P.V MFMAs + V reads  8 tile barrier  16 staging  64 max exchange
#   (bits 4 = Q.K MFMAs + K reads and 32 = bias reads existed until the far-tile path restructured that phase; their figures — 17.6 / 19.4
#   and 1.2 / 5.6 us of 92 / 90 — are in DESIGN.md 4.2)
MASKS="1 2 8 16 64 3 24 27"
cd "$(dirname "$0")/.."
if [ "$1" != run ]; then
  for m in $MASKS; do
    ( mkdir -p music2midi_amd/csrc/build_awcut$m
      hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DM2M_AW_CUT=$m -c music2midi_amd/csrc/enc_kernels.hip -o music2midi_amd/csrc/build_awcut$m/enc_kernels.o 2>/dev/null
      objs=$(ls music2midi_amd/csrc/build/*.o | grep -v enc_kernels.o)
      hipcc -shared -fPIC --offload-arch=gfx950 -o music2midi_amd/lib/libmusic2midi_amd_awcut$m.so $objs music2midi_amd/csrc/build_awcut$m/enc_kernels.o && echo built $m ) &
    while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
  done
  wait
  exit 0
fi
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in 0 $MASKS; do
  L=$PWD/music2midi_amd/lib/libmusic2midi_amd${m:+_awcut$m}.so; [ $m = 0 ] && L=$PWD/music2midi_amd/lib/libmusic2midi_amd.so
  D=gpurun_out/awcut_$m; rm -rf $D
  OCC_BATCHES=4,12,32 M2M_LIBRARY=$L rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python3 tools/attn_occupancy.py > $D.log 2>&1
  python3 - $D/t_kernel_trace.csv $m <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "attn_wide" in r["Kernel_Name"]: acc[int(r["Grid_Size_X"] if "Grid_Size_X" in r else r["Grid_Size"]) // 256].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(f"cut {int(sys.argv[2]):3d}: " + "   ".join(f"{g:5d} workgroups {sum(v) / len(v) / 1e3:7.2f} us" for g, v in sorted(acc.items())))
PY
  rm -rf $D
done
