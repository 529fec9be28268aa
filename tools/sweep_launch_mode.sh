#!/bin/bash
# pairs per launch x launch groups in flight on the current build (bench.py --batch / --streams, driver flags otherwise): one line per point
cd "$(dirname "$0")/.."
for streams in 1 2 3; do for batch in 1 2 4 5 10 20; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-measurements --streams $streams --batch $batch 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('groups $streams  pairs/launch $batch  ', round(d['value'],1), 'pairs/s', [round(x,1) for x in d['value_repeats']['pairs_per_s']])"
done; done
