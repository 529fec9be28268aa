#!/bin/bash
# several library builds on ONE box, alternating: tools/ab_bench_n.sh <rounds> <lib> [<lib> ...]
cd "$(dirname "$0")/.."
N=$1; shift
for i in $(seq 1 $N); do for L in "$@"; do
  ICEMATCH_LIB=$L timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-measurements 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d.get('kernel_ms_per_pair',{})
print('$L', round(d['value'],2), 'pairs/s', round(d['ms_per_step'],3), 'ms', {x:k[x] for x in k if 'gemm' in x and 'lg_' in x})"
done; done
