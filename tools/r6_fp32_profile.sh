#!/bin/bash
# Round 6: the DEFAULT (fp32, bit-exact) mode of the Python classes under the profiler - what evaluate.py gets.
#   bash tools/r6_fp32_profile.sh <tag>   -> gpurun_out/prof_<tag>/{fp32_bench.json, fp32_kernel_stats.csv, fp32_pmc_traffic_summary.txt}
TAG=${1:-r6_fp32}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/prof_$TAG
mkdir -p $D
python3 bench.py --precision fp32 --no-parity --no-native --no-frontend --no-train --cpu-tokens 0 > $D/fp32_bench.json 2> $D/fp32_bench.err
echo "bench done"; cut -c1-600 $D/fp32_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $D/ks -o ks -- python3 bench.py --precision fp32 --steps 3 --warmup 1 --no-roofline --no-parity --no-native --no-frontend --no-train --cpu-tokens 0 > $D/ks.log 2>&1
cp $D/ks/ks_kernel_stats.csv $D/fp32_kernel_stats.csv 2>/dev/null
export PMC_OUT=$D/pmc_traffic PMC_PRECISION=fp32; bash tools/pmc_traffic.sh > $D/pmc_traffic.log 2>&1; cp $PMC_OUT/summary.txt $D/fp32_pmc_traffic_summary.txt 2>/dev/null
rm -rf $D/ks $D/pmc_traffic/*.csv
head -12 $D/fp32_kernel_stats.csv | cut -c1-160; cat $D/fp32_pmc_traffic_summary.txt | cut -c1-220
