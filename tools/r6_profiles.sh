#!/bin/bash
# Round-6 profile set (the round-5 set + the two new ones below), run on the GPU box from the repo root:  bash tools/r5_profiles.sh <tag>   (writes gpurun_out/prof_<tag>/)
#   bench.json                       the default bench line of this build
#   kernel_stats.csv                 rocprofv3 --kernel-trace --stats of the headline bench command (decode kernels)
#   pmc_traffic_summary.txt          FETCH_SIZE / WRITE_SIZE per decode-kernel launch, separate passes, WITH the clips/launch header
#   pmc_mfma_summary.txt             matrix-core counters of the inference kernels
#   frontend_*                       configs[1]: log-mel kernel stats + traffic + issue-side counters (VALU / LDS instructions, LDS bank conflicts, wait cycles)
#   train_dropout_kernel_stats.csv   configs[4] share: 23 directly issued 16-clip steps with dropout 0.1 (tools/train_gap.py)
#   train_pmc_l1_l2_summary.txt      per kernel of that step: L2 hit rate, L1 accesses per cycle and CU, matrix-core busy
#   train_fp8_dropout_kernel_stats.csv   the same step in the MXFP8 mode (per-product A/B)
#   encoder_kernel_stats.csv / encoder_two_kernel_path_kernel_stats.csv   tools/enc_bench.py: fused norm+GEMM vs rmsnorm + gemm
#   native_{before,after}_kernel_stats.csv, native_pmc_l2_summary.txt     tools/r6_native_profile.sh: 128 x 3 s chunk, (clip, head) workgroups vs 4 clips per workgroup
#   fp32_bench.json, fp32_kernel_stats.csv, fp32_pmc_traffic_summary.txt  tools/r6_fp32_profile.sh: the default (bit-exact) mode
TAG=${1:-r6}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/prof_$TAG
mkdir -p $D
python3 bench.py > $D/bench.json 2> $D/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $D/ks -o ks -- python3 bench.py --steps 3 --warmup 1 --no-roofline --no-parity --no-native --no-frontend --no-train --cpu-tokens 0 > $D/ks.log 2>&1
cp $D/ks/ks_kernel_stats.csv $D/kernel_stats.csv 2>/dev/null
echo "kernel stats done"
export PMC_OUT=$D/pmc_traffic; bash tools/pmc_traffic.sh > $D/pmc_traffic.log 2>&1; cp $PMC_OUT/summary.txt $D/pmc_traffic_summary.txt 2>/dev/null
export PMC_OUT=$D/pmc_mfma; bash tools/pmc_mfma.sh > $D/pmc_mfma.log 2>&1; cp $PMC_OUT/summary.txt $D/pmc_mfma_summary.txt 2>/dev/null
unset PMC_OUT
echo "pmc done"
rocprofv3 --kernel-trace --stats --output-format csv -d $D/fe -o fe -- python3 tools/frontend_bench.py 64 > $D/fe.log 2>&1
cp $D/fe/fe_kernel_stats.csv $D/frontend_kernel_stats.csv 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D/fe_pmc -o $C -- python3 tools/frontend_bench.py 64 > $D/fe_$C.log 2>&1
done
python3 - "$D" <<'PY' > $D/frontend_pmc_traffic_summary.txt
import csv, glob, sys
d = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{d}/fe_pmc/{c}_counter_collection.csv")
    if not f: print("missing", c); continue
    n, tot = 0, 0.0
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c and "logmel" in r["Kernel_Name"]:
            n += 1; tot += float(r["Counter_Value"])
    corr = 2.0 if c == "FETCH_SIZE" else 1.0
    print(f"logmel kernel B=64  {c}/launch {tot / max(n, 1):10.1f} KiB over {n} launches  ({'x2 corrected ' if corr == 2 else ''}{corr * tot / max(n, 1) * 1024 / 1e6:8.2f} MB)")
PY
export PMC_OUT=$D/pmc_frontend_valu; bash tools/pmc_frontend_valu.sh > $D/pmc_frontend_valu.log 2>&1; cp $PMC_OUT/summary.txt $D/frontend_pmc_valu_summary.txt 2>/dev/null; rm -f $PMC_OUT/*.csv; unset PMC_OUT
echo "frontend done"
M2M_GAP_DROPOUT=0.1 M2M_TRAIN_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $D/tr -o tr -- python3 tools/train_gap.py > $D/tr.log 2>&1
cp $D/tr/tr_kernel_stats.csv $D/train_dropout_kernel_stats.csv 2>/dev/null
grep WALL $D/tr.log
# round 5: the same step with the projection products on MXFP8 (per-product A/B against the bf16 step above)
M2M_GAP_PREC=fp8 M2M_GAP_DROPOUT=0.1 M2M_TRAIN_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $D/tr8 -o tr8 -- python3 tools/train_gap.py > $D/tr8.log 2>&1
cp $D/tr8/tr8_kernel_stats.csv $D/train_fp8_dropout_kernel_stats.csv 2>/dev/null
grep WALL $D/tr8.log
# round 5: encoder + cross-K/V alone at the headline geometry, fused norm+GEMM (default) and the two-kernel path
rocprofv3 --kernel-trace --stats --output-format csv -d $D/enc -o enc -- python3 tools/enc_bench.py > $D/enc.log 2>&1
cp $D/enc/enc_kernel_stats.csv $D/encoder_kernel_stats.csv 2>/dev/null
M2M_NORM_GEMM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $D/enc0 -o enc0 -- python3 tools/enc_bench.py > $D/enc0.log 2>&1
cp $D/enc0/enc0_kernel_stats.csv $D/encoder_two_kernel_path_kernel_stats.csv 2>/dev/null
grep bf16 $D/enc.log $D/enc0.log
M2M_GAP_DROPOUT=0.1 bash tools/pmc_train_l2.sh > $D/train_pmc_l1_l2_summary.txt 2> $D/pmc_train.err
rm -rf $D/ks $D/fe $D/tr $D/tr8 $D/enc $D/enc0 $D/fe_pmc $D/pmc_traffic/*.csv $D/pmc_mfma/*.csv gpurun_out/pmc_train_l2/*.csv 2>/dev/null
# round 6: the reference's own chunk before / after the large-chain kernels (kernel stats + L2 read counters), and the DEFAULT (fp32) mode
bash tools/r6_native_profile.sh ${TAG}_native > $D/native.log 2>&1
cp gpurun_out/prof_${TAG}_native/native_*_kernel_stats.csv gpurun_out/prof_${TAG}_native/native_pmc_l2_summary.txt $D/ 2>/dev/null
bash tools/r6_fp32_profile.sh ${TAG}_fp32 > $D/fp32.log 2>&1
cp gpurun_out/prof_${TAG}_fp32/fp32_bench.json gpurun_out/prof_${TAG}_fp32/fp32_kernel_stats.csv gpurun_out/prof_${TAG}_fp32/fp32_pmc_traffic_summary.txt $D/ 2>/dev/null
ls -la $D; head -3 $D/pmc_traffic_summary.txt | cut -c1-200; cat $D/frontend_pmc_traffic_summary.txt; head -12 $D/train_pmc_l1_l2_summary.txt | cut -c1-230
