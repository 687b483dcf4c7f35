#!/bin/bash
# L2 hit rate + duration of dw_group_kernel under a given environment:  [M2M_DW_XCD=1] bash tools/pmc_dw_l2.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_dw_l2_$1
mkdir -p $OUT
M2M_TRAIN_GRAPH=0 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT -o p -- python3 tools/train_gap.py > $OUT/p.log 2>&1 || tail -3 $OUT/p.log
python3 - $OUT <<'PY'
import csv, collections, sys
agg = collections.defaultdict(lambda: [0, 0.0]); dur = []
for r in csv.DictReader(open(sys.argv[1] + "/p_counter_collection.csv")):
    if "dw_group_kernel" not in r["Kernel_Name"]: continue
    a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    if r["Counter_Name"] == "TCC_HIT_sum": dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
print(sys.argv[1], "  ".join(f"{c}={v[1] / v[0]:.4g}" for c, v in sorted(agg.items())), f"avg duration {sum(dur) / len(dur) / 1e3:.1f} us")
PY
