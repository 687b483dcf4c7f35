mkdir -p gpurun_out/r2p
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_end_to_end_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r2p/pytest.log; tail -5 gpurun_out/r2p/pytest.log
timeout -k 10 200 python tools/train_bench.py bf16 > gpurun_out/r2p/train.txt 2>&1
timeout -k 10 200 python tools/train_bench.py fp8 >> gpurun_out/r2p/train.txt 2>&1
grep -v "^/opt" gpurun_out/r2p/train.txt
hipcc --offload-arch=gfx950 -O3 tools/xcd_barrier.hip -o /tmp/xcd_barrier 2> gpurun_out/r2p/cc.log && timeout -k 10 240 /tmp/xcd_barrier > gpurun_out/r2p/xcd_barrier.txt 2>&1; echo "xcd rc $?"
grep -c STALE gpurun_out/r2p/xcd_barrier.txt; grep "ld.sc1 " gpurun_out/r2p/xcd_barrier.txt | head -12
