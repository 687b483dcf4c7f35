mkdir -p gpurun_out/r2j
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2j/tg -o tg -- python3 tools/train_gap.py > gpurun_out/r2j/tg.log 2>&1; grep WALL gpurun_out/r2j/tg.log
cp gpurun_out/r2j/tg/tg_kernel_stats.csv gpurun_out/r2j/train_gap_kernel_stats.csv; rm -rf gpurun_out/r2j/tg
