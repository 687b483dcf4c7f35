cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof_drop
M2M_GAP_DROPOUT=0.1 M2M_TRAIN_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_drop/t -o t -- python3 tools/train_gap.py > gpurun_out/prof_drop/run.log 2>&1
cp gpurun_out/prof_drop/t/t_kernel_stats.csv gpurun_out/prof_drop/train_dropout_kernel_stats.csv
grep WALL gpurun_out/prof_drop/run.log
rm -rf gpurun_out/prof_drop/t
