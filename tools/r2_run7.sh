mkdir -p gpurun_out/r2g
hipcc --offload-arch=gfx950 -O3 tools/xcd_barrier.hip -o /tmp/xcd_barrier 2> gpurun_out/r2g/cc.log && timeout -k 10 240 /tmp/xcd_barrier > gpurun_out/r2g/xcd_barrier.txt 2>&1; echo "xcd rc $?"
hipcc --offload-arch=gfx950 -O3 tools/barrier_floor.hip -o /tmp/barrier_floor 2>> gpurun_out/r2g/cc.log && timeout -k 10 240 /tmp/barrier_floor > gpurun_out/r2g/barrier_floor.txt 2>&1; echo "floor rc $?"
cat gpurun_out/r2g/xcd_barrier.txt | head -80
