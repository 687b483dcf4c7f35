mkdir -p gpurun_out/r2y
for m in fwd bwd; do echo "== fuse $m" ; M2M_TRAIN_GRAPH=0 M2M_TRAIN_FUSE_PV=$m timeout 300 python tools/_dbg_dropout.py 2>&1 | grep -v "^/opt" | head -8; done > gpurun_out/r2y/dbg2.txt
cat gpurun_out/r2y/dbg2.txt
