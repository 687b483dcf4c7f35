#!/bin/bash
# timeline of ONE training step (16 clips, S = 261, 256 labels, dropout 0.1, direct issue): every dispatch with start / duration / gap to
# the previous END on the device, from rocprofv3's kernel trace:   bash tools/r3_step_timeline.sh [ENV=...]   -> gpurun_out/timeline/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/timeline
env "$@" M2M_GAP_DROPOUT=0.1 M2M_TRAIN_GRAPH=0 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/timeline/d -o t -- python3 tools/train_gap.py > gpurun_out/timeline/run.log 2>&1
grep WALL gpurun_out/timeline/run.log
cp $(find gpurun_out/timeline/d -name "*kernel_trace.csv" | head -1) gpurun_out/timeline/trace.csv
rm -rf gpurun_out/timeline/d
python3 - <<'PY'
import csv, re
rows=list(csv.DictReader(open('gpurun_out/timeline/trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[re.sub(r'\(.*','',r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').replace('m2m::','')) for r in rows]
# steps start at train_prologue_kernel; take the last complete one
starts=[i for i,n in enumerate(names) if n.startswith('train_prologue_kernel')]
a,b=starts[-2],starts[-1]
t0=int(rows[a]['Start_Timestamp'])
out=open('gpurun_out/timeline/step.txt','w')
prev_end=t0; busy_end=t0; gaps=0; 
print(f"{'#':>4s} {'start us':>9s} {'dur us':>8s} {'gap us':>7s}  kernel (gap = start - latest end so far; negative = overlaps)", file=out)
for i in range(a,b):
    s=int(rows[i]['Start_Timestamp']); e=int(rows[i]['End_Timestamp'])
    gap=(s-busy_end)/1e3
    if gap>0: gaps+=gap
    print(f"{i-a:4d} {(s-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} {gap:7.1f}  q{rows[i]['Queue_Id']} {names[i][:90]} grid={rows[i]['Grid_Size_X']}x{rows[i]['Grid_Size_Y']}x{rows[i]['Grid_Size_Z']} wg={rows[i]['Workgroup_Size_X']}", file=out)
    busy_end=max(busy_end,e)
print(f"step span {(busy_end-t0)/1e3:.1f} us, idle gaps {gaps:.1f} us, dispatches {b-a}", file=out)
out.close()
print(open('gpurun_out/timeline/step.txt').read()[-300:])
PY
