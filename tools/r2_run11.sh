mkdir -p gpurun_out/r2k
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_mx8_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r2k/pytest.log; tail -3 gpurun_out/r2k/pytest.log
timeout -k 10 200 python tools/train_bench.py bf16 > gpurun_out/r2k/train.txt 2>&1
timeout -k 10 200 python tools/train_bench.py fp8 >> gpurun_out/r2k/train.txt 2>&1
grep -v "^/opt" gpurun_out/r2k/train.txt
