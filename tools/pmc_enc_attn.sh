#!/bin/bash
# Issue-side counters of the encoder flash attention kernel (tools/enc_bench.py, B = 32, S = 864).  bash tools/pmc_enc_attn.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${PMC_OUT:-gpurun_out/pmc_enc_attn}
mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
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
        if not any(t in k for t in ("attn_kernel<m2m::bf16_t", "attn_wide", "norm_gemm", "gemm_kernel<m2m::bf16_t", "resid_panel")): continue
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in sorted(agg.items()):
    print(k[:70])
    for c, v in sorted(d.items()):
        print(f"    {c:32s} per launch {v[1] / v[0]:16.1f}   ({v[0]} launches)")
PY
rm -f $OUT/*.csv
