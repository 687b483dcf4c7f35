#!/bin/bash
# re-collect the decode-side parts of the r4 profile set (kernel stats, PMC traffic, matrix-core counters) into gpurun_out/prof_<tag>/
TAG=${1:-r4_a}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/prof_$TAG
mkdir -p $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D/ks -o ks -- python3 bench.py --steps 3 --warmup 1 --no-roofline --no-parity --no-native --no-frontend --no-train --cpu-tokens 0 > $D/ks.log 2>&1
cp $D/ks/ks_kernel_stats.csv $D/kernel_stats.csv 2>/dev/null
export PMC_OUT=$D/pmc_traffic; bash tools/pmc_traffic.sh > $D/pmc_traffic.log 2>&1; cp $PMC_OUT/summary.txt $D/pmc_traffic_summary.txt 2>/dev/null
export PMC_OUT=$D/pmc_mfma; bash tools/pmc_mfma.sh > $D/pmc_mfma.log 2>&1; cp $PMC_OUT/summary.txt $D/pmc_mfma_summary.txt 2>/dev/null
rm -rf $D/ks $D/pmc_traffic/*.csv $D/pmc_mfma/*.csv
cat $D/pmc_traffic_summary.txt | cut -c1-220; head -12 $D/pmc_mfma_summary.txt | cut -c1-200; head -12 $D/kernel_stats.csv | cut -c1-200
