#!/bin/bash
# A/B of the fused feed-forward kernel against the separate-launch form (one gpurun call):
#   gpurun --timeout 1500 -- 'bash tools/ab_ffn.sh'
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/ab_ffn
mkdir -p $out
cd $root
python3 -m pytest tests -q -x -m gpu > $out/pytest.log 2>&1
tail -3 $out/pytest.log
for mode in fused unfused; do
  if [ $mode = unfused ]; then export IM_FFN_UNFUSED=1; else unset IM_FFN_UNFUSED; fi
  python3 bench.py --config 5 --no-cpu-baseline > $out/bench5_$mode.json 2> $out/bench5_$mode.err
  python3 - <<PY
import json
b = json.load(open("$out/bench5_$mode.json"))
k = b.get("kernel_ms_per_pair", {})
print("$mode config5", round(b["value"], 3), "pairs/s", {n: v for n, v in k.items() if "mlp" in n or "ffn" in n})
PY
done
