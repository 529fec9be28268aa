#!/bin/bash
# round 6 dev loop of the feed-forward kernel: the kernel tests, the bench line, the in-kernel timeline
cd "$(dirname "$0")/.."
O=gpurun_out/ffn_dev
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16_stress.py -m gpu -x -q -k "ffn" > $O/pytest_ffn.log 2>&1
tail -5 $O/pytest_ffn.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-measurements > $O/bench.json 2> $O/bench.err
python - $O/bench.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
k = d.get("kernel_ms_per_pair", {})
print("value", d["value"], "ms_per_step", d["ms_per_step"], {x: k[x] for x in k if "ffn" in x or "attn" in x or "gemm" in x})
P
bash tools/ffn_stamps.sh 2>/dev/null
