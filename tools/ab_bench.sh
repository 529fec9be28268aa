#!/bin/bash
# A/B of two library builds on ONE box, alternating: tools/ab_bench.sh <libA> <libB> [rounds] [bench flags...]
cd "$(dirname "$0")/.."
A=$1; B=$2; N=${3:-2}; shift 3
for i in $(seq 1 $N); do for L in $A $B; do
  ICEMATCH_LIB=$L timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-measurements "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d.get('kernel_ms_per_pair',{})
print('$L', round(d['value'],2), 'pairs/s', round(d['ms_per_step'],3), 'ms', {x:k[x] for x in k if 'ffn' in x})"
done; done
