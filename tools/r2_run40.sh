mkdir -p gpurun_out/r2w
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
sed 's/precision="bf16"/precision="fp8"/' tools/train_gap.py > /tmp/train_gap_fp8.py; cp /tmp/train_gap_fp8.py tools/_train_gap_fp8.py
M2M_TRAIN_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2w/tg -o tg -- python3 tools/_train_gap_fp8.py > gpurun_out/r2w/tg.log 2>&1; grep WALL gpurun_out/r2w/tg.log
cp gpurun_out/r2w/tg/tg_kernel_stats.csv gpurun_out/r2w/fp8_kernel_stats.csv; rm -rf gpurun_out/r2w/tg
M2M_FP8_PARTS=fwd,dx timeout 200 python tools/train_bench.py fp8 2>&1 | grep -v "^/opt" > gpurun_out/r2w/fp8_fwd_dx.txt; timeout 200 python tools/train_bench.py fp8 2>&1 | grep -v "^/opt" > gpurun_out/r2w/fp8_all.txt; cat gpurun_out/r2w/fp8_fwd_dx.txt gpurun_out/r2w/fp8_all.txt
