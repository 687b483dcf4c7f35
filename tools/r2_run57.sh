timeout 300 python tools/frontend_bench.py 64 2>&1 | grep -v "^/opt"
timeout 600 python -m pytest tests/test_frontend_gpu.py tests/test_golden_gpu.py -x -q -k "logmel or frontend" 2>&1 | tail -3
