#!/bin/bash
# per-kernel durations of the two projection forms at ten pairs per launch, one launch group (rocprofv3 kernel trace of bench.py --streams 1 --batch 10)
cd "$(dirname "$0")/.."
O=gpurun_out/proj_stats; mkdir -p $O; export TMPDIR=/tmp
for v in 1 0; do
  rm -rf $O/t$v
  IM_PROJ_TILED=$v rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$v -- python3 bench.py --no-cpu-baseline --no-side-measurements --streams 1 --batch 10 --steps 20 --warmup 5 > $O/bench_$v.json 2>/dev/null
  python3 - $O/t$v $v <<'P'
import csv, glob, sys, statistics, collections
fn = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
d = collections.defaultdict(list)
for r in csv.DictReader(open(fn)):
    n = r["Kernel_Name"]
    if "proj_rows" in n or "gemm_nt_kernel<128, 128, 32, 3" in n or "gemm_nt_kernel<128, 128, 32, 5" in n or "ffn_fused_split" in n:
        d[n[:70]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items():
    big = [x for x in v if x > 0.5 * max(v)]
    print("IM_PROJ_TILED=" + sys.argv[2], k, "launches", len(v), "median of the full-width launches %.1f us" % (statistics.median(big) / 1e3))
P
  rm -rf $O/t$v
done
