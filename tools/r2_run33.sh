mkdir -p gpurun_out/r2v
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_golden_gpu.py tests/test_t5_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r2v/pytest.log; tail -3 gpurun_out/r2v/pytest.log
timeout -k 10 200 python tools/train_bench.py bf16 > gpurun_out/r2v/train.txt 2>&1
grep -v "^/opt" gpurun_out/r2v/train.txt
