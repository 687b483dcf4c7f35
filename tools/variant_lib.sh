#!/bin/bash
# Build lib/libmusic2midi_amd_<tag>.so from the product objects with ONE source recompiled under extra flags (same-box A/Bs of a
# compile-time switch without a full tagged build):  tools/variant_lib.sh <tag> <source stem, e.g. enc_kernels> <flags...>
cd "$(dirname "$0")/.."
tag=$1; stem=$2; shift 2
mkdir -p music2midi_amd/csrc/build_$tag
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c music2midi_amd/csrc/$stem.hip -o music2midi_amd/csrc/build_$tag/$stem.o || exit 1
objs=$(ls music2midi_amd/csrc/build/*.o | grep -v "/$stem.o")
hipcc -shared -fPIC --offload-arch=gfx950 -o music2midi_amd/lib/libmusic2midi_amd_$tag.so $objs music2midi_amd/csrc/build_$tag/$stem.o && echo "built music2midi_amd/lib/libmusic2midi_amd_$tag.so"
