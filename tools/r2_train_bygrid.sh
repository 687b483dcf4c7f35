mkdir -p gpurun_out/bygrid
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
M2M_TRAIN_GRAPH=0 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bygrid/t -o t -- python3 tools/train_gap.py > gpurun_out/bygrid/run.log 2>&1
python tools/trace_by_grid.py gpurun_out/bygrid/t/t_kernel_trace.csv > gpurun_out/bygrid/by_grid.txt
rm -rf gpurun_out/bygrid/t
grep WALL gpurun_out/bygrid/run.log
