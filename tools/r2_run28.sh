mkdir -p gpurun_out/r2u
echo "== fp8 all" > gpurun_out/r2u/train.txt; timeout -k 10 200 python tools/train_bench.py fp8 >> gpurun_out/r2u/train.txt 2>&1
echo "== fp8 fwd,dx (dW bf16 grouped)" >> gpurun_out/r2u/train.txt; M2M_FP8_PARTS=fwd,dx timeout -k 10 200 python tools/train_bench.py fp8 >> gpurun_out/r2u/train.txt 2>&1
echo "== fp8 fwd only" >> gpurun_out/r2u/train.txt; M2M_FP8_PARTS=fwd timeout -k 10 200 python tools/train_bench.py fp8 >> gpurun_out/r2u/train.txt 2>&1
grep -v "^/opt" gpurun_out/r2u/train.txt
