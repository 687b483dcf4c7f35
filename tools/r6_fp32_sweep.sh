set -u
L=$PWD/music2midi_amd/lib
mkdir -p gpurun_out/r6i
run() { echo "== $*"; env "$@" python tools/native_mc_sweep.py 32 220500 fp32 1024 ${SPEC:-0,0,0} 2>&1 | tee -a gpurun_out/r6i/fp32_sweep.log; }
run A=0
for t in pc3 pc4 ps3 pb3; do run M2M_LIBRARY=$L/libmusic2midi_amd_$t.so; done
run A=0
for r in 0 2 3 6; do run M2M_KV_RESIDENT_LAYERS=$r; done
SPEC="0,0,32 0,0,11 0,0,8" run A=0
