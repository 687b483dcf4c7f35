#!/bin/bash
# Round 6: the reference's own chunk (128 x 3 s segments, S = 190, bf16) under the profiler, BEFORE (one (clip, head) per
# attention workgroup, one hidden slice per feed-forward workgroup: M2M_DA_CLIPS=1 M2M_DEC_FF_SLICES=1) and AFTER (the defaults by chain size:
# 4 clips per workgroup, 4 slices).   bash tools/r6_native_profile.sh <tag>  ->  gpurun_out/prof_<tag>/
#   native_{before,after}_kernel_stats.csv    rocprofv3 --kernel-trace --stats, 2 x 1 023 decode steps
#   native_pmc_l2_summary.txt                 per decode kernel and launch: L1 -> L2 read requests (TCP_TCC_READ_REQ: what a workgroup
#                                             pulls from L2, weights included) against the bytes that came from beyond L2 (FETCH_SIZE, x2)
TAG=${1:-r6_native}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/prof_$TAG
mkdir -p $D
for leg in before after; do
  if [ $leg = before ]; then export M2M_DA_CLIPS=1 M2M_DEC_FF_SLICES=1; else unset M2M_DA_CLIPS M2M_DEC_FF_SLICES; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/ks_$leg -o ks -- python3 tools/native_prof.py 128 1024 > $D/ks_$leg.log 2>&1
  cp $D/ks_$leg/ks_kernel_stats.csv $D/native_${leg}_kernel_stats.csv 2>/dev/null
  i=0
  for G in "TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $G --output-format csv -d $D/pmc_$leg -o p$i -- python3 tools/native_prof.py 128 129 > $D/pmc_${leg}_$i.log 2>&1 || tail -3 $D/pmc_${leg}_$i.log
  done
done
python3 - $D <<'PY' > $D/native_pmc_l2_summary.txt
import csv, collections, glob, sys
d = sys.argv[1]
print("# reference-native chunk: 128 x S = 190, bf16, two chains of 64 clips; per decode-kernel LAUNCH (one chain = 64 clips), 128 decode steps")
print("# L2 reads = TCP_TCC_READ_REQ x 64 B (what the CUs request from L2: K/V stream + weights + rows); beyond L2 = FETCH_SIZE x2 (gfx950) ; separate --pmc passes")
for leg in ("before", "after"):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in sorted(glob.glob(f"{d}/pmc_{leg}/p*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "dec_" not in k: continue
            a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    print(f"== {leg}" + ("  (M2M_DA_CLIPS=1 M2M_DEC_FF_SLICES=1: rounds 1-5)" if leg == "before" else "  (defaults by chain size: dec_attn_mc_kernel C = 4, dec_ff_multi_kernel 4 slices)"))
    tot_l2 = tot_hbm = 0.0
    for k in sorted(agg):
        c = {n: v[1] / max(v[0], 1) for n, v in agg[k].items()}
        n = max(v[0] for v in agg[k].values())
        l2 = c.get("TCP_TCC_READ_REQ_sum", 0) * 64 / 1e6
        hbm = 2 * c.get("FETCH_SIZE", 0) * 1024 / 1e6
        wr = c.get("WRITE_SIZE", 0) * 1024 / 1e6
        tot_l2 += l2 * n; tot_hbm += hbm * n
        print(f"{k[:66]:66s} launches {n:5d}  L2 reads/launch {l2:8.2f} MB  beyond L2 {hbm:8.2f} MB  written {wr:6.2f} MB  L2 reads / beyond-L2 {l2 / max(hbm, 1e-9):5.2f}")
    print(f"   all decode kernels: L2 reads {tot_l2 / 1e3:8.2f} GB, beyond L2 {tot_hbm / 1e3:8.2f} GB over the run -> ratio {tot_l2 / max(tot_hbm, 1e-9):4.2f}")
PY
rm -rf $D/ks_before $D/ks_after $D/pmc_before $D/pmc_after
head -8 $D/native_before_kernel_stats.csv | cut -c1-150; head -8 $D/native_after_kernel_stats.csv | cut -c1-150; cat $D/native_pmc_l2_summary.txt | cut -c1-230
