#!/bin/bash
# What bounds the grouped weight-gradient launch (dw_group_kernel, 16-clip training step)?  Separate PMC passes (kernel trace only).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${PMC_OUT:-gpurun_out/pmc_dw}
mkdir -p $OUT
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  M2M_TRAIN_GRAPH=0 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT -o p$i -- python3 tools/train_gap.py > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
done
python3 - <<'PY'
import csv, collections, glob, os
out = os.environ.get("PMC_OUT", "gpurun_out/pmc_dw")
for f in sorted(glob.glob(f"{out}/p*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "dw_group_kernel" not in r["Kernel_Name"]: continue
        a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    print(os.path.basename(f), "  ".join(f"{c}={v[1] / v[0]:.4g}" for c, v in sorted(agg.items())))
PY
