#!/bin/bash
# round 6 dev loop of the feed-forward kernel: the kernel test (both forms, bit-identical), then the bench line with 32 and 64 rows per block
cd "$(dirname "$0")/.."
O=gpurun_out/ffn_dev
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "ffn" > $O/pytest_ffn.log 2>&1
tail -5 $O/pytest_ffn.log
for rows in 32 64; do
  IM_FFN_ROWS=$rows timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_rows$rows.json 2> $O/bench_rows$rows.err
  python - $O/bench_rows$rows.json $rows <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
k = d.get("kernel_ms_per_pair", {})
print("rows", sys.argv[2], "value", d["value"], "ms_per_step", d["ms_per_step"], {x: k[x] for x in k if "ffn" in x or "attn" in x or "gemm" in x})
P
done
