mkdir -p gpurun_out/train_check
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_train_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/train_check/pytest.log; tail -3 gpurun_out/train_check/pytest.log
timeout -k 10 200 python tools/train_bench.py bf16 > gpurun_out/train_check/train.txt 2>&1
grep -v "^/opt" gpurun_out/train_check/train.txt
M2M_TRAIN_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/train_check/tg -o tg -- python3 tools/train_gap.py > gpurun_out/train_check/tg.log 2>&1; grep WALL gpurun_out/train_check/tg.log
cp gpurun_out/train_check/tg/tg_kernel_stats.csv gpurun_out/train_check/train_gap_kernel_stats.csv; rm -rf gpurun_out/train_check/tg
grep "attn_stripe" gpurun_out/train_check/train_gap_kernel_stats.csv | cut -c1-160
