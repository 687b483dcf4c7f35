mkdir -p gpurun_out/r2l
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_mx8_gpu.py tests/test_end_to_end_gpu.py tests/test_callers_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r2l/pytest.log; tail -5 gpurun_out/r2l/pytest.log
echo "== graph+side" > gpurun_out/r2l/train.txt; timeout -k 10 200 python tools/train_bench.py bf16 >> gpurun_out/r2l/train.txt 2>&1
echo "== side, no graph" >> gpurun_out/r2l/train.txt; M2M_TRAIN_GRAPH=0 timeout -k 10 200 python tools/train_bench.py bf16 >> gpurun_out/r2l/train.txt 2>&1
echo "== single stream" >> gpurun_out/r2l/train.txt; M2M_TRAIN_SIDE=0 timeout -k 10 200 python tools/train_bench.py bf16 >> gpurun_out/r2l/train.txt 2>&1
echo "== fp8 graph+side" >> gpurun_out/r2l/train.txt; timeout -k 10 200 python tools/train_bench.py fp8 >> gpurun_out/r2l/train.txt 2>&1
grep -v "^/opt" gpurun_out/r2l/train.txt
