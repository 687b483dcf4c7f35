mkdir -p gpurun_out/r2n
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_mx8_gpu.py tests/test_end_to_end_gpu.py tests/test_callers_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r2n/pytest.log; tail -5 gpurun_out/r2n/pytest.log
echo "== grouped dW + graph" > gpurun_out/r2n/train.txt; timeout -k 10 200 python tools/train_bench.py bf16 >> gpurun_out/r2n/train.txt 2>&1
echo "== grouped dW, no graph" >> gpurun_out/r2n/train.txt; M2M_TRAIN_GRAPH=0 timeout -k 10 200 python tools/train_bench.py bf16 >> gpurun_out/r2n/train.txt 2>&1
echo "== side stream (no group)" >> gpurun_out/r2n/train.txt; M2M_TRAIN_DW_GROUP=0 timeout -k 10 200 python tools/train_bench.py bf16 >> gpurun_out/r2n/train.txt 2>&1
echo "== fp32" >> gpurun_out/r2n/train.txt; timeout -k 10 200 python tools/train_bench.py fp32 >> gpurun_out/r2n/train.txt 2>&1
echo "== fp8" >> gpurun_out/r2n/train.txt; timeout -k 10 200 python tools/train_bench.py fp8 >> gpurun_out/r2n/train.txt 2>&1
grep -v "^/opt" gpurun_out/r2n/train.txt
