#!/bin/bash
# in-kernel timeline of the fused feed-forward, both forms at 1 / 2 / 10 pairs per launch (diagnostic build with shader-clock stamps)
cd "$(dirname "$0")/.."
bash tools/build_ffn_variant.sh f_STAMP -DIM_FSTAMP > /dev/null
for split in 0 1; do for pairs in 1 2 10; do
ICEMATCH_LIB=build_abl/f_STAMP/libicematch.so IM_FFN_SPLIT=$split PAIRS=$pairs timeout 300 python tools/ffn_stamps.py
done; done
