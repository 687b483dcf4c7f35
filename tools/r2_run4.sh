mkdir -p gpurun_out/r2d
timeout 1200 python -m pytest tests/test_train_gpu.py -x -q -s 2>&1 | tail -30 > gpurun_out/r2d/train.log
tail -14 gpurun_out/r2d/train.log
python tools/train_bench.py bf16 2>&1 | grep -E "^bf16|rror"
python tools/train_bench.py fp32 2>&1 | grep -E "^fp32|rror"
