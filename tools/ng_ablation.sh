for m in "" ngs1 ngs2 ngs4 ngs8 ngs3 ngs7; do
  if [ -z "$m" ]; then L=$PWD/music2midi_amd/lib/libmusic2midi_amd.so; else L=$PWD/music2midi_amd/lib/libmusic2midi_amd_$m.so; fi
  echo "== ${m:-product}: $(M2M_LIBRARY=$L python tools/enc_bench.py 2>&1 | grep bf16)"
done
