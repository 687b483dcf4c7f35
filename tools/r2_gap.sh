cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/gap; mkdir -p gpurun_out/gap
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gap -o g -- python3 tools/train_gap.py > gpurun_out/gap/log.txt 2>&1
grep WALL gpurun_out/gap/log.txt
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/gap/g_kernel_trace.csv")))
tot=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows)
print("kernels", len(rows), "per step", len(rows)/23, "busy ms per step", tot/23/1e6)
small=sum(1 for r in rows if int(r["End_Timestamp"])-int(r["Start_Timestamp"])<8000)
print("kernels under 8 us:", small/23, "per step")
PY
rm -f gpurun_out/gap/g_kernel_trace.csv
