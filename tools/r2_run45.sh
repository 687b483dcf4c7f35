mkdir -p gpurun_out/r2y
echo "== default" > gpurun_out/r2y/dbg.txt; timeout 300 python tools/_dbg_dropout.py 2>&1 | grep -v "^/opt" | head -40 >> gpurun_out/r2y/dbg.txt
echo "== no graph" >> gpurun_out/r2y/dbg.txt; M2M_TRAIN_GRAPH=0 timeout 300 python tools/_dbg_dropout.py 2>&1 | grep -v "^/opt" | head -40 >> gpurun_out/r2y/dbg.txt
echo "== no fuse pv" >> gpurun_out/r2y/dbg.txt; M2M_TRAIN_FUSE_PV=0 timeout 300 python tools/_dbg_dropout.py 2>&1 | grep -v "^/opt" | head -20 >> gpurun_out/r2y/dbg.txt
echo "== no stripes" >> gpurun_out/r2y/dbg.txt; M2M_TRAIN_STRIPES=0 timeout 300 python tools/_dbg_dropout.py 2>&1 | grep -v "^/opt" | head -20 >> gpurun_out/r2y/dbg.txt
cat gpurun_out/r2y/dbg.txt
