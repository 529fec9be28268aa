#!/bin/bash
# Builds an A/B variant of libicematch.so into build_abl/<name>/ (travels with gpurun; select it with ICEMATCH_LIB=...):
#   tools/build_variant.sh <name> [-DFLAG ...]           current sources with extra compiler flags
#   tools/build_variant.sh <name> --rev <git-rev> file.hip [file.hip ...]   the named files taken from a git revision
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
out=$root/build_abl/$name
mkdir -p $out/src; ln -sfn $root/include $root/build_abl/include
cp $root/icepy4d_amd/csrc/*.hip $root/icepy4d_amd/csrc/*.h $out/src/
flags=()
if [ "$1" = "--rev" ]; then
  rev=$2; shift 2
  for f in "$@"; do git -C $root show $rev:icepy4d_amd/csrc/$f > $out/src/$f; done
else
  flags=("$@")
fi
cd $out/src
rm -f *.o
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$root/include -I$root/icepy4d_amd/csrc/../../include "${flags[@]}" -c $f -o ${f%.hip}.o &
done
wait
for f in *.hip; do [ -f ${f%.hip}.o ] || { echo "compiling $f FAILED"; exit 1; }; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libicematch.so *.o
echo built $out/libicematch.so
