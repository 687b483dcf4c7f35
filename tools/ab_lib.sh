#!/bin/bash
# same-box A/B of the product library against a tagged build on the encoder bench (kernel averages under rocprofv3, two rounds)
#   tools/ab_lib.sh <tag> [kernel-name filter, default: every kernel above 1 %]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; flt=${2:-.}
for t in "" $tag "" $tag; do
  L=$PWD/music2midi_amd/lib/libmusic2midi_amd${t:+_$t}.so
  D=gpurun_out/ab_${t:-product}
  rm -rf $D; M2M_LIBRARY=$L rocprofv3 --kernel-trace --stats --output-format csv -d $D -o enc -- python3 tools/enc_bench.py > $D.log 2>&1
  echo "== ${t:-product}: $(grep bf16 $D.log)"
  python3 - $D/enc_kernel_stats.csv "$flt" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if not any(t in r["Name"] for t in ("bf16", "wide", "resid_panel", "norm_gemm")): continue
    if re.search(sys.argv[2], r["Name"]) and float(r["Percentage"]) > 0.3:
        print(f"   {r['Name'][:72]:72s} n {r['Calls']:>4s}  avg {float(r['AverageNs']) / 1000:8.2f} us")
PY
done
