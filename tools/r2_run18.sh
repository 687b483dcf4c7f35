mkdir -p gpurun_out/r2q
timeout -k 10 600 python -m pytest tests/test_rccl_gpu.py -x -q -s 2>&1 | tail -15 > gpurun_out/r2q/pytest.log; cat gpurun_out/r2q/pytest.log
MASTER_ADDR=127.0.0.1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2 --warmup 1 --no-parity --no-train --no-frontend --cpu-tokens 0 > gpurun_out/r2q/torchrun_bench.json 2> gpurun_out/r2q/torchrun_bench.err; echo rc $?; cut -c1-300 gpurun_out/r2q/torchrun_bench.json; tail -3 gpurun_out/r2q/torchrun_bench.err
