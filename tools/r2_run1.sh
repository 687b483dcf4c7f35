set -x
mkdir -p gpurun_out/r2a
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
timeout 600 python bench.py > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err; tail -3 gpurun_out/r2a/bench.err; cat gpurun_out/r2a/bench.json | cut -c1-3000
for lib in "" c3s2 c4s2 c5s2 c6s2 c4s3; do
  for q in "" 8; do
    ( [ -n "$lib" ] && export M2M_LIBRARY=$PWD/music2midi_amd/lib/libmusic2midi_amd_$lib.so; [ -n "$q" ] && export GPU_MAX_HW_QUEUES=$q; timeout 300 python tools/chain_sweep.py 32 2>&1 | grep rows/chain ) >> gpurun_out/r2a/sweep.log
  done
done
cat gpurun_out/r2a/sweep.log
