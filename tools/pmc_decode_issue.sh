#!/bin/bash
# Issue-side counters of the decode kernels (the greedy loop at B = 32, 32 steps).  bash tools/pmc_decode_issue.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${PMC_OUT:-gpurun_out/pmc_decode_issue}
mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o p$i -- python3 bench.py --steps 1 --warmup 0 --no-roofline --no-parity --no-native --no-frontend --no-train --cpu-tokens 0 --max-length 33 > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
done
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, collections, glob, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in sorted(glob.glob(f"{out}/p*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "dec_" not in k: continue
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in sorted(agg.items()):
    print(k[:70])
    for c, v in sorted(d.items()):
        print(f"    {c:28s} per launch {v[1] / v[0]:14.1f}   ({v[0]} launches)")
PY
rm -f $OUT/*.csv
