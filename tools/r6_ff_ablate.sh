OUT=gpurun_out/r6s; mkdir -p $OUT; export TMPDIR=/tmp
L=$PWD/music2midi_amd/lib
for t in product ffa1 ffa2 ffa4 ffa7; do
  lib=$L/libmusic2midi_amd_$t.so; [ $t = product ] && lib=$L/libmusic2midi_amd.so
  M2M_LIBRARY=$lib rocprofv3 --kernel-trace -d $OUT/$t -o np -- python3 tools/native_prof.py 128 512 > $OUT/$t.log 2>&1
  echo "== $t"; python3 tools/kernel_stats.py $OUT/$t 4; rm -rf $OUT/$t
done
