#!/bin/bash
# A/B variant of libicematch.so that differs in ffn_fused.hip only: compiled with the given flags, linked with the in-tree objects.
#   tools/build_ffn_variant.sh <name> [-DFLAG ...]   ->  build_abl/<name>/libicematch.so   (select with ICEMATCH_LIB=...)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
out=$root/build_abl/$name
mkdir -p $out
cd $root/icepy4d_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$root/include "$@" -c ffn_fused.hip -o $out/ffn_fused.o 2>/dev/null
objs=$(ls *.o | grep -v '^ffn_fused.o$')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libicematch.so $objs $out/ffn_fused.o
echo built $out/libicematch.so
