#!/bin/bash
# HBM traffic of the decode kernels from PMC counters, collected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (4 TCC slots per pass), no trace domains
# other than the kernel trace.  Run on the GPU box from the repo root:  bash tools/pmc_traffic.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${PMC_OUT:-gpurun_out/pmc_traffic}
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o $C -- \
    python3 bench.py --precision ${PMC_PRECISION:-bf16} --steps 1 --warmup 0 --no-roofline --no-parity --no-native --no-frontend --no-train --cpu-tokens 0 --max-length 49 > $OUT/$C.log 2>&1 || tail -5 $OUT/$C.log
done
python3 - <<'PY'
import csv, collections, glob
import os
out = os.environ.get("PMC_OUT", "gpurun_out/pmc_traffic")
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{c}_counter_collection.csv")
    if not f:
        print("missing", c, glob.glob(out + "/*")); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0]
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    res[c] = agg
# clips per decode-kernel launch of the profiled run (bench.py reads this header and refuses a summary without it): the bench
# decodes 32 clips per GPU as chains of M2M_GROUP_ROWS clips (default: two chains of 16 from 24 clips on, DESIGN_HISTORY.md 4.4)
batch = int(os.environ.get("PMC_BATCH", "32"))
rows = int(os.environ.get("M2M_GROUP_ROWS") or (16 if batch >= 24 else batch))
with open(out + "/summary.txt", "w") as fh:
    head = f"# decode kernels: clips/launch = {min(rows, batch)}   (batch {batch} per GPU, precision {os.environ.get('PMC_PRECISION', 'bf16')}; FETCH_SIZE x2-corrected for gfx950; separate --pmc passes)"
    print(head); fh.write(head + "\n")
    for k in sorted(set(res.get("FETCH_SIZE", {})) | set(res.get("WRITE_SIZE", {}))):
        if "dec_" not in k and "logmel" not in k and "gemm_kernel" not in k and "attn_kernel" not in k and "attn_wide" not in k and "resid_panel" not in k: continue
        n, fs = res.get("FETCH_SIZE", {}).get(k, [0, 0.0]); _, ws = res.get("WRITE_SIZE", {}).get(k, [0, 0.0])
        n = max(n, 1)
        # counters are in KiB; gfx950 FETCH_SIZE counts wide coalesced reads at 1/2 -> x2 (MI355X_MICROARCH.md HBM)
        line = f"{k[:70]:70s} launches {n:6d}  FETCH_SIZE/launch {fs / n:10.1f} KiB (x2 corrected {2 * fs / n * 1024 / 1e6:8.2f} MB)  WRITE_SIZE/launch {ws / n:9.1f} KiB ({ws / n * 1024 / 1e6:6.3f} MB)"
        print(line); fh.write(line + "\n")
PY
