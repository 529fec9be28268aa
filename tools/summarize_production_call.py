#!/usr/bin/env python3
"""Sum of the kernels of the ONE measured call of tools/run_production_call.py from a rocprofv3 kernel trace: everything between the last two launches of the
marker (erfinv) kernel. Usage: summarize_production_call.py <rocprof dir> <stdout json of the run> <out csv>"""
import csv, glob, json, os, sys
d, run_json, out = sys.argv[1:4]
tr = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(tr)))
name = lambda r: r.get("Kernel_Name") or r.get("kernel_name")
st = lambda r: int(r.get("Start_Timestamp") or r.get("start_timestamp"))
en = lambda r: int(r.get("End_Timestamp") or r.get("end_timestamp"))
rows.sort(key=st)
marks = [i for i, r in enumerate(rows) if "erfinv" in name(r).lower()]
i0, i1 = marks[-2], marks[-1]
seg = rows[i0 + 1:i1]
agg = {}
for r in seg:
    n = name(r)
    n = n if len(n) < 120 else n[:117] + "..."
    a = agg.setdefault(n, [0, 0])
    a[0] += 1; a[1] += en(r) - st(r)
tot = sum(v[1] for v in agg.values())
span = en(seg[-1]) - st(seg[0])
# busy time = union of kernel intervals (kernels of different streams overlap)
busy, cur_s, cur_e = 0, None, None
for r in seg:
    s, e = st(r), en(r)
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
run = json.loads(open(run_json).read().strip().split("\n")[-1])
with open(out, "w") as f:
    f.write(f"# one production call (tools/run_production_call.py): wall {run['wall_ms']} ms; kernels launched inside it: {len(seg)}; sum of kernel durations {tot / 1e6:.2f} ms; "
            f"GPU busy (union of kernel intervals) {busy / 1e6:.2f} ms; first kernel start to last kernel end {span / 1e6:.2f} ms\n")
    f.write(f"# timer split (ms): {json.dumps(run['timer_split_ms'])}; matched points {run['matched_points']}\n")
    for l in run["tile_pairs_log"]:
        f.write(f"# {l}\n")
    f.write("kernel,calls,total_ms,percent_of_kernel_time\n")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write(f"\"{n}\",{c},{t / 1e6:.3f},{100 * t / tot:.2f}\n")
print(open(out).read()[:3000])
