mkdir -p gpurun_out/r2r
timeout -k 10 900 python -m pytest tests/test_train_gpu.py -x -q -s -k "graph_replay or dropout" 2>&1 | tail -25 > gpurun_out/r2r/pytest.log; cat gpurun_out/r2r/pytest.log | grep -v "^$" | tail -12
