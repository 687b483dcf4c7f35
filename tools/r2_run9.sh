mkdir -p gpurun_out/r2i
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_golden_gpu.py tests/test_t5_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r2i/pytest.log; tail -3 gpurun_out/r2i/pytest.log
echo "== new default" > gpurun_out/r2i/train.txt; timeout -k 10 200 python tools/train_bench.py bf16 >> gpurun_out/r2i/train.txt 2>&1
echo "== old dW + big tiles only" >> gpurun_out/r2i/train.txt; M2M_TRAIN_DW_OLD=1 M2M_GEMM_SMALL_BELOW=0 timeout -k 10 200 python tools/train_bench.py bf16 >> gpurun_out/r2i/train.txt 2>&1
echo "== old dW, small tiles" >> gpurun_out/r2i/train.txt; M2M_TRAIN_DW_OLD=1 timeout -k 10 200 python tools/train_bench.py bf16 >> gpurun_out/r2i/train.txt 2>&1
for w in 256 384 768; do echo "== DW_WGS $w" >> gpurun_out/r2i/train.txt; M2M_DW_WGS=$w timeout -k 10 200 python tools/train_bench.py bf16 >> gpurun_out/r2i/train.txt 2>&1; done
for t in 64 128; do echo "== DW_TILE $t" >> gpurun_out/r2i/train.txt; M2M_DW_TILE=$t timeout -k 10 200 python tools/train_bench.py bf16 >> gpurun_out/r2i/train.txt 2>&1; done
echo "== DW_TILE 64 WGS 256" >> gpurun_out/r2i/train.txt; M2M_DW_TILE=64 M2M_DW_WGS=256 timeout -k 10 200 python tools/train_bench.py bf16 >> gpurun_out/r2i/train.txt 2>&1
echo "== small below 1024" >> gpurun_out/r2i/train.txt; M2M_GEMM_SMALL_BELOW=1024 timeout -k 10 200 python tools/train_bench.py bf16 >> gpurun_out/r2i/train.txt 2>&1
echo "== fp32" >> gpurun_out/r2i/train.txt; timeout -k 10 200 python tools/train_bench.py fp32 >> gpurun_out/r2i/train.txt 2>&1
grep -v "^/opt" gpurun_out/r2i/train.txt
