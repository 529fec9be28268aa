#!/bin/bash
# timing ablations of the attention's main loop (attention_bx.hip, -DBX_ABL_*; WRONG results by construction): stage entry at n = 4096, self attention,
# 1 / 2 / 4 / 10 pairs per launch (tools/bench_attn_batch.py), one line per build
cd "$(dirname "$0")/.."
for v in "" NO_VREAD NO_KREAD "NO_VREAD NO_KREAD" NO_EXPCUT NO_STAGE NO_BARRIER NO_MFMA "NO_VREAD NO_KREAD NO_EXPCUT" "NO_VREAD NO_KREAD NO_EXPCUT NO_STAGE NO_BARRIER"; do
  name=a_$(echo ${v:-ASIS} | tr ' ' '_')
  flags=""; for f in $v; do flags="$flags -DBX_ABL_$f"; done
  bash tools/build_variant_file.sh $name attention_bx.hip $flags > /dev/null
  echo "== ${v:-as is}"
  ICEMATCH_LIB=build_abl/$name/libicematch.so timeout 300 python tools/bench_attn_batch.py 4096 2>/dev/null | grep "cross=0 bf16" | sed 's/TFLOP.*//' 
done
