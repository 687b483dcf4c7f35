mkdir -p gpurun_out/mxs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/mxs/t -o t -- python3 tools/mx_gemm_shapes.py 20 > gpurun_out/mxs/run.log 2>&1
python tools/trace_by_grid.py gpurun_out/mxs/t/t_kernel_trace.csv mxgemm > gpurun_out/mxs/by_grid.txt
cat gpurun_out/mxs/by_grid.txt
rm -rf gpurun_out/mxs/t
