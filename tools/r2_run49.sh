mkdir -p gpurun_out/r2z
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_mx8_gpu.py -x -q -k "fp8 or mx" 2>&1 | tail -5
echo "== fused q"; timeout 200 python tools/train_bench.py fp8 2>&1 | grep -v "^/opt"
echo "== separate quantisers"; M2M_FP8_FUSED_Q=0 timeout 200 python tools/train_bench.py fp8 2>&1 | grep -v "^/opt"
