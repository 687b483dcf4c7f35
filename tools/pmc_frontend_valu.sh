#!/bin/bash
# Issue-side counters of the log-mel kernel (BASELINE configs[1], 64 clips): VALU / LDS / SALU instruction counts and the busy
# fraction of the VALU, to show what the kernel is bound by (it is far from the HBM roofline by design of the measurement:
# 26 flop/B).  Separate passes per counter group (kernel trace only).  Run on the GPU box:  bash tools/pmc_frontend_valu.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${PMC_OUT:-gpurun_out/pmc_frontend_valu}
mkdir -p $OUT
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_SCA"; do
  N=$(echo $G | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT -o $N -- python3 tools/frontend_bench.py 64 > $OUT/$N.log 2>&1 || tail -3 $OUT/$N.log
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("PMC_OUT", "gpurun_out/pmc_frontend_valu")
agg = collections.defaultdict(float); n = collections.Counter(); dur = 0.0; nd = 0
for f in glob.glob(f"{out}/*_counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        if "logmel" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        if r["Dispatch_Id"] not in seen and r["Counter_Name"].startswith("SQ_INSTS_VALU"):
            seen.add(r["Dispatch_Id"]); dur += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); nd += 1
lines = []
for k in sorted(agg):
    lines.append(f"logmel kernel B=64  {k:24s} per launch {agg[k] / max(n[k], 1):16.1f}   ({n[k]} launches)")
frames = 64 * 862
if "SQ_INSTS_VALU" in agg:
    v = agg["SQ_INSTS_VALU"] / n["SQ_INSTS_VALU"]
    lines.append(f"VALU wave-instructions per frame: {v / frames:.0f}; LDS {agg.get('SQ_INSTS_LDS', 0) / max(n['SQ_INSTS_LDS'], 1) / frames:.0f}; SALU {agg.get('SQ_INSTS_SALU', 0) / max(n['SQ_INSTS_SALU'], 1) / frames:.0f}")
    if nd:
        us = dur / nd / 1e3
        lines.append(f"kernel {us:.1f} us under the profiler; VALU issue time at 4 cycles per wave-instruction on 1024 SIMDs @ 2.4 GHz: {v * 4 / 1024 / 2.4e3:.1f} us = {100 * v * 4 / 1024 / 2.4e3 / us:.0f} % of the kernel")
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
