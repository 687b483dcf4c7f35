mkdir -p gpurun_out/r2e
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r2e/pytest.log; tail -4 gpurun_out/r2e/pytest.log
timeout 900 python bench.py > gpurun_out/r2e/bench.json 2> gpurun_out/r2e/bench.err; tail -2 gpurun_out/r2e/bench.err; cut -c1-1200 gpurun_out/r2e/bench.json
