mkdir -p gpurun_out/r2b
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > gpurun_out/r2b/pytest.log
tail -15 gpurun_out/r2b/pytest.log
