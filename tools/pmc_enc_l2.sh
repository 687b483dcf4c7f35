#!/bin/bash
# L2 behaviour of the encoder kernels (tools/enc_bench.py, B = 32, S = 864).  bash tools/pmc_enc_l2.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${PMC_OUT:-gpurun_out/pmc_enc_l2}
mkdir -p $OUT
i=0
for C in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o p$i -- python3 tools/enc_bench.py > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
done
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, collections, glob, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in sorted(glob.glob(f"{out}/p*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "bf16_t" not in k and "norm_gemm" not in k and "resid_panel" not in k: continue
        if not any(t in k for t in ("attn_kernel", "norm_gemm", "gemm_kernel", "resid_panel")): continue
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in sorted(agg.items()):
    print(k[:70])
    for c, v in sorted(d.items()):
        print(f"    {c:32s} per launch {v[1] / v[0]:16.1f}   ({v[0]} launches)")
PY
rm -f $OUT/*.csv
