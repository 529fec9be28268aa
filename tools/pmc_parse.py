#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 `--pmc ... --output-format csv` counter_collection file:
    python tools/pmc_parse.py <counter_collection.csv> <kernel name substring> [...]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
first = None
for r in rows:
    k = r["Kernel_Name"][:44]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if first is None: first = r["Counter_Name"]
    if r["Counter_Name"] == first: cnt[k] += 1
for k, v in agg.items():
    if any(s in k for s in sys.argv[2:]):
        n = cnt[k]
        print(k, {c: round(x / n) for c, x in v.items()})
