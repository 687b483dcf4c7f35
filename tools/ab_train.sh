#!/bin/bash
# same-box A/B of the product library against a tagged build on the directly issued training step (tools/train_gap.py under rocprofv3):
#   tools/ab_train.sh <tag> [kernel-name filter]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; flt=${2:-attn_head}
for t in "" $tag "" $tag; do
  L=$PWD/music2midi_amd/lib/libmusic2midi_amd${t:+_$t}.so
  D=gpurun_out/abt_${t:-product}
  rm -rf $D; M2M_LIBRARY=$L M2M_GAP_DROPOUT=0.1 M2M_TRAIN_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o tr -- python3 tools/train_gap.py > $D.log 2>&1
  echo "== ${t:-product}: $(grep WALL $D.log)"
  python3 - $D/tr_kernel_stats.csv "$flt" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]): print(f"   {r['Name'][:72]:72s} n {r['Calls']:>4s}  avg {float(r['AverageNs']) / 1000:8.2f} us")
PY
done
