#!/bin/bash
# A/B variant of libicematch.so that differs in ONE source file: compiled with the given flags, linked with the in-tree objects.
#   tools/build_variant_file.sh <name> <file.hip> [-DFLAG ...]   ->  build_abl/<name>/libicematch.so   (select with ICEMATCH_LIB=...)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; file=$2; shift 2
base=$(basename $file .hip)
out=$root/build_abl/$name
mkdir -p $out
cd $root/icepy4d_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$root/include "$@" -c $base.hip -o $out/$base.o 2>/dev/null
objs=$(ls *.o | grep -v "^$base.o\$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libicematch.so $objs $out/$base.o
echo built $out/libicematch.so
