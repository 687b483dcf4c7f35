#!/bin/bash
# per-kernel totals of one training step (16 clips, S = 261, 256 labels, dropout 0.1, direct issue) for two settings, side by side:
#   bash tools/r3_step_stats.sh "ENV_A=..." "ENV_B=..."        (results: gpurun_out/stepstats/)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/stepstats
i=0
for S in "$@"; do
  env $S M2M_GAP_DROPOUT=0.1 M2M_TRAIN_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stepstats/d$i -o t -- python3 tools/train_gap.py > gpurun_out/stepstats/run$i.log 2>&1
  grep WALL gpurun_out/stepstats/run$i.log
  cp $(find gpurun_out/stepstats/d$i -name "*kernel_stats.csv" | head -1) gpurun_out/stepstats/stats$i.csv
  rm -rf gpurun_out/stepstats/d$i
  i=$((i+1))
done
python3 - "$@" <<'PY'
import csv, sys, re
def load(f):
    d={}
    for r in csv.DictReader(open(f)):
        n=re.sub(r'\(.*','',r['Name']).replace('void ','').replace('m2m::','')
        d[n]=(int(r['Calls'])/23.0, float(r['TotalDurationNs'])/23e3, float(r['AverageNs'])/1e3)
    return d
a=load('gpurun_out/stepstats/stats0.csv'); b=load('gpurun_out/stepstats/stats1.csv') if len(sys.argv)>2 else {}
names=sorted(set(a)|set(b), key=lambda n:-(a.get(n,(0,0,0))[1]+b.get(n,(0,0,0))[1]))
ta=sum(v[1] for v in a.values()); tb=sum(v[1] for v in b.values())
print(f"{'kernel':70s} {'A: n/step':>9s} {'us/step':>8s} {'avg':>6s} | {'B: n/step':>9s} {'us/step':>8s} {'avg':>6s}")
for n in names[:48]:
    x=a.get(n,(0,0,0)); y=b.get(n,(0,0,0))
    print(f"{n[:70]:70s} {x[0]:9.1f} {x[1]:8.1f} {x[2]:6.1f} | {y[0]:9.1f} {y[1]:8.1f} {y[2]:6.1f}")
print(f"{'TOTAL kernel us per step':70s} {sum(v[0] for v in a.values()):9.1f} {ta:8.1f}        | {sum(v[0] for v in b.values()):9.1f} {tb:8.1f}")
PY
