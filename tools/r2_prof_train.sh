cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2_train_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_train_prof -o tr -- python3 tools/train_bench.py bf16 > gpurun_out/r2_train_prof/log.txt 2>&1
ls gpurun_out/r2_train_prof
head -30 gpurun_out/r2_train_prof/tr_kernel_stats.csv
