mkdir -p gpurun_out/r2y
timeout 300 python tools/_dbg_dropout.py 2>&1 | grep -v "^/opt" | head -8 > gpurun_out/r2y/dbg3.txt; cat gpurun_out/r2y/dbg3.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_end_to_end_gpu.py tests/test_callers_gpu.py -x -q 2>&1 | tail -5
timeout -k 10 200 python tools/train_bench.py bf16 2>&1 | grep -v "^/opt"
