#!/bin/bash
# Matrix-core utilisation of the encoder / decode kernels from PMC counters (one pass, kernel trace only):
#   MfmaUtil  = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * 1024 SIMDs)  (rocprofv3's own formula)
#   TFLOP/s   = SQ_INSTS_VALU_MFMA_MOPS_BF16 * 512 / kernel duration                 (MOPS = math ops / 512)
# Run on the GPU box from the repo root:  bash tools/pmc_mfma.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${PMC_OUT:-gpurun_out/pmc_mfma}
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $OUT -o mfma -- \
  python3 bench.py --steps 1 --warmup 0 --no-roofline --no-parity --no-native --no-frontend --no-train --cpu-tokens 0 --max-length 33 > $OUT/mfma.log 2>&1 || tail -5 $OUT/mfma.log
python3 - <<'PY'
import csv, collections, glob
import os
out = os.environ.get("PMC_OUT", "gpurun_out/pmc_mfma")
f = glob.glob(f"{out}/mfma_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); dur = collections.defaultdict(float)
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    did = r["Dispatch_Id"]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if did not in seen:
        seen.add(did); calls[k] += 1; dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
XCDS = 8   # the csv carries GRBM_GUI_ACTIVE summed over the 8 XCD instances; the formula takes their max
lines = []
for k in sorted(agg, key=lambda k: -dur[k]):
    if not any(t in k for t in ("gemm_kernel", "attn_kernel", "attn_wide", "resid_panel", "dec_ff", "logmel")): continue
    a = agg[k]; n = calls[k]
    busy, gui, mops = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), a.get("GRBM_GUI_ACTIVE", 0.0), a.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
    util = 100.0 * busy / (gui / XCDS * 1024) if gui else 0.0
    tf = mops * 512 / dur[k] / 1e3 if dur[k] else 0.0     # flops per ns -> TFLOP/s
    lines.append(f"{k[:64]:64s} launches {n:5d}  avg {dur[k] / n / 1e3:9.2f} us  MfmaUtil {util:6.2f} %  bf16 MFMA {tf:8.1f} TFLOP/s ({100 * tf / 2500:5.2f} % of 2.5 PF)")
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
