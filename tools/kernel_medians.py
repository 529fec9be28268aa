#!/usr/bin/env python3
"""Median duration per kernel of a rocprofv3 --kernel-trace output directory (csv):  python tools/kernel_medians.py DIR [substring ...]"""
import collections, csv, glob, statistics, sys
d = collections.defaultdict(list)
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        d[r["Kernel_Name"].replace("void ", "").split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if len(sys.argv) > 2 and not any(s in k for s in sys.argv[2:]):
        continue
    print(f"{k[:60]:60s} n={len(v):4d} median {statistics.median(v):9.2f} us  min {min(v):9.2f}")
