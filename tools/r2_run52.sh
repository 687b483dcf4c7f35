timeout -k 10 900 python -m pytest tests/test_mx8_gpu.py -x -q 2>&1 | tail -6
