mkdir -p gpurun_out/r2s
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_end_to_end_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r2s/pytest.log; tail -5 gpurun_out/r2s/pytest.log
echo "== stripes" > gpurun_out/r2s/train.txt; timeout -k 10 200 python tools/train_bench.py bf16 >> gpurun_out/r2s/train.txt 2>&1
echo "== no stripes" >> gpurun_out/r2s/train.txt; M2M_TRAIN_STRIPES=0 timeout -k 10 200 python tools/train_bench.py bf16 >> gpurun_out/r2s/train.txt 2>&1
echo "== fp32 stripes" >> gpurun_out/r2s/train.txt; timeout -k 10 200 python tools/train_bench.py fp32 >> gpurun_out/r2s/train.txt 2>&1
grep -v "^/opt" gpurun_out/r2s/train.txt
