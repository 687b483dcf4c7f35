timeout -k 10 900 python -m pytest tests/test_mx8_gpu.py tests/test_train_gpu.py -x -q -k "mx or fp8" 2>&1 | tail -4
echo "== wide from 512"; timeout 200 python tools/train_bench.py fp8 2>&1 | grep -v "^/opt"
echo "== wide from 384"; M2M_MXQ_WIDE_FROM=384 timeout 200 python tools/train_bench.py fp8 2>&1 | grep -v "^/opt"
echo "== never wide"; M2M_MXQ_WIDE_FROM=100000 timeout 200 python tools/train_bench.py fp8 2>&1 | grep -v "^/opt"
