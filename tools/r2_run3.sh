mkdir -p gpurun_out/r2c
timeout 1200 python -m pytest tests/test_train_gpu.py -x -q -s 2>&1 | tail -60 > gpurun_out/r2c/train.log
tail -40 gpurun_out/r2c/train.log
