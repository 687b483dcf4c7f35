mkdir -p gpurun_out/r2h
SWEEP_ROWS=16,12,11,8 timeout -k 10 300 python tools/chain_sweep.py 32 > gpurun_out/r2h/sweep_dflt.txt 2>&1
GPU_MAX_HW_QUEUES=8 SWEEP_ROWS=16,12,11,8 timeout -k 10 300 python tools/chain_sweep.py 32 > gpurun_out/r2h/sweep_q8.txt 2>&1
GPU_MAX_HW_QUEUES=6 SWEEP_ROWS=16,11 timeout -k 10 300 python tools/chain_sweep.py 32 > gpurun_out/r2h/sweep_q6.txt 2>&1
grep rows/chain gpurun_out/r2h/*.txt
