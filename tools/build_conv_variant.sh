#!/bin/bash
# A/B variant of libicematch.so that differs in conv_wino.hip / conv_wino_bx2.hip only: compiled with the given flags, linked with the in-tree objects.
#   tools/build_conv_variant.sh <name> [-DFLAG ...]   ->  build_abl/<name>/libicematch.so   (select with ICEMATCH_LIB=...)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
out=$root/build_abl/$name
mkdir -p $out
cd $root/icepy4d_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -c conv_wino.hip -o $out/conv_wino.o 2>/dev/null &
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -c conv_wino_bx2.hip -o $out/conv_wino_bx2.o 2>/dev/null &
wait
objs=$(ls *.o | grep -v '^conv_wino.o$' | grep -v '^conv_wino_bx2.o$')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libicematch.so $objs $out/conv_wino.o $out/conv_wino_bx2.o
echo built $out/libicematch.so
