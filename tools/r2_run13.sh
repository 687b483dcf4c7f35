mkdir -p gpurun_out/r2m
hipcc --offload-arch=gfx950 -O3 tools/stream_rate.hip -o /tmp/stream_rate 2> gpurun_out/r2m/cc.log && timeout -k 10 300 /tmp/stream_rate > gpurun_out/r2m/stream_rate.txt 2>&1; echo rc $?
cat gpurun_out/r2m/stream_rate.txt
