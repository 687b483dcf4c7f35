mkdir -p gpurun_out/r2f
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r2f/pytest.log; tail -4 gpurun_out/r2f/pytest.log
timeout 900 python bench.py > gpurun_out/r2f/bench.json 2> gpurun_out/r2f/bench.err; tail -2 gpurun_out/r2f/bench.err; cut -c1-600 gpurun_out/r2f/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2f/tg -o tg -- python3 tools/train_gap.py > gpurun_out/r2f/tg.log 2>&1; tail -3 gpurun_out/r2f/tg.log
cp gpurun_out/r2f/tg/tg_kernel_stats.csv gpurun_out/r2f/train_gap_kernel_stats.csv; rm -rf gpurun_out/r2f/tg
