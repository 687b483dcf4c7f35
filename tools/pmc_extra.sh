#!/bin/bash
# extra PMC passes for the decode kernels (L2 hit rate, wave stall mix).  bash tools/pmc_extra.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_extra
mkdir -p $OUT
i=0
for C in "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o p$i -- \
    python3 bench.py --steps 1 --warmup 0 --no-roofline --no-parity --no-native --no-frontend --no-train --cpu-tokens 0 --max-length 33 > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
done
python3 - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob("gpurun_out/pmc_extra/p*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "dec_" not in k: continue
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, d in sorted(agg.items()):
        print(k[:60].ljust(60), "  ".join(f"{c}={v[1] / v[0]:.0f}" for c, v in sorted(d.items())))
PY
