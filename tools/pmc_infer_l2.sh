#!/bin/bash
# Per kernel of the headline generate pass (32 clips, 256 decode steps): L2 hits / misses, L1 accesses, average duration (separate PMC passes, kernel trace only).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_infer_l2
mkdir -p $OUT
i=0
for G in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT -o p$i -- python3 bench.py --steps 1 --warmup 0 --no-roofline --no-parity --no-native --no-frontend --no-train --cpu-tokens 0 --max-length 257 > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
done
python3 - $OUT <<'PY'
import csv, collections, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0])); dur = collections.defaultdict(lambda: [0, 0.0])
for f in sorted(glob.glob(sys.argv[1] + "/p*_counter_collection.csv")):
    first = None
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:58] + " g" + r["Grid_Size"]
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
        if first is None: first = r["Counter_Name"]
        if r["Counter_Name"] == first and f.endswith("p1_counter_collection.csv"):
            d = dur[k]; d[0] += 1; d[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
rows = sorted(agg, key=lambda k: -dur[k][1])
for k in rows[:26]:
    d = dur[k]
    if not d[0]: continue
    c = {n: v[1] / v[0] for n, v in agg[k].items()}
    hit = c.get("TCC_HIT_sum", 0); miss = c.get("TCC_MISS_sum", 0)
    print(f"{k:78s} n={d[0]:4d} {d[1] / d[0] / 1e3:7.1f} us  L2 hit {100 * hit / max(hit + miss, 1):4.0f}%  miss {miss:9.3g}  L1 acc {c.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0):9.3g}"
          f"  acc/cycle/CU {c.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0) / 256 / max(d[1] / d[0] * 2.4, 1):5.2f}  VALU {c.get('SQ_INSTS_VALU', 0):9.3g}  mfma busy {100 * c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(d[1] / d[0] * 2.4 * 1024, 1):4.0f}%")
PY
