mkdir -p gpurun_out/r2x
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_mx8_gpu.py -x -q -s -k "fp8 or mx" 2>&1 | tail -25 > gpurun_out/r2x/pytest.log; grep -v "^$" gpurun_out/r2x/pytest.log | tail -8
timeout 200 python tools/train_bench.py fp8 2>&1 | grep -v "^/opt"
