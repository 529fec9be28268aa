"""Condenses the rocprofv3 outputs of tools/collect_profiles.sh (gpurun_out/<tag>/) into the small files kept under profiles/:
per-kernel duration tables (median / mean over the launches of the kernel trace) and PMC-derived HBM traffic per launch."""
import collections, csv, glob, json, os, statistics, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:70]


def newest(fns):
    """gpurun MERGES a call's outputs into gpurun_out/: a second collection of the same tag leaves two traces side by side. Only the
    newest file of a directory is the current build's."""
    fns = sorted(fns, key=os.path.getmtime)
    return fns[-1:]


def durations(sub):
    fns = newest(glob.glob(f"{src}/{sub}/**/*kernel_trace.csv", recursive=True))
    d = collections.defaultdict(list)
    for fn in fns:
        for r in csv.DictReader(open(fn)):
            d[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return d


for sub, outname in (("stats1", f"{tag}_kernel_stats_streams1_batch1.csv"), ("stats10", f"{tag}_kernel_stats_streams1_batch10.csv"),
                     ("stats_default", f"{tag}_kernel_stats_default.csv"),
                     ("stats_config5", f"{tag}_kernel_stats_config5.csv"), ("stats_assign", f"{tag}_kernel_stats_assign_one_pair.csv")):
    d = durations(sub)
    if not d:
        continue
    tot = sum(sum(v) for v in d.values())
    with open(os.path.join(dst, outname), "w") as fh:
        fh.write("kernel,calls,median_us,mean_us,min_us,max_us,total_ms,percent\n")
        for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
            if len(v) < 3:
                continue
            fh.write(f"\"{k}\",{len(v)},{statistics.median(v)/1e3:.2f},{statistics.mean(v)/1e3:.2f},{min(v)/1e3:.2f},{max(v)/1e3:.2f},"
                     f"{sum(v)/1e6:.3f},{100*sum(v)/tot:.2f}\n")


def counters(sub):
    agg = collections.defaultdict(list)
    for fn in newest(glob.glob(f"{src}/{sub}/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(fn)):
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


traffic = {"_comment": "HBM-side traffic per launch from separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of "
                       "tools/run_pair_once.py (one 1080p / 4096-keypoint LightGlue pair; one 12 MP / 16384-keypoint SuperGlue pair, "
                       "keys suffixed @16384). bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE under-counts wide coalesced "
                       "reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section; calibrated there for 16-byte-per-lane streaming), "
                       "WRITE_SIZE is exact. Medians over the launches of each kernel."}
for mode, suffix in (("lg", ""), ("sg", "@16384")):
    f, w = counters(f"pmc_{mode}_FETCH_SIZE"), counters(f"pmc_{mode}_WRITE_SIZE")
    for k in sorted(set(f) | set(w)):
        if not (k.startswith("im::") or "im::" in k):
            continue
        fk = statistics.median(f[k]) if k in f else 0.0
        wk = statistics.median(w[k]) if k in w else 0.0
        traffic[k + suffix] = {"launches": len(f.get(k, w.get(k, []))), "FETCH_SIZE_KB": round(fk, 1), "WRITE_SIZE_KB": round(wk, 1),
                               "traffic_bytes": int((2 * fk + wk) * 1024)}
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
sq = counters("pmc_attn_sq")
for fn in newest(glob.glob(f"{src}/pmc_attn_sq/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        if "flash_attn" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(os.path.join(dst, f"{tag}_attn_pmc_sq.csv"), "w") as fh:
        fh.write("counter,mean,launches\n")
        for k, v in agg.items():
            fh.write(f"{k},{sum(v)/len(v):.1f},{len(v)}\n")
# matrix-pipe busy share of every kernel of one pair: SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) over SQ_BUSY_CYCLES
# (32 instances per launch, one per shader engine: / 32 = busy cycles of the launch)
for fn in newest(glob.glob(f"{src}/pmc_sq_all/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fn)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = []
    for k, c in agg.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        if m.get("SQ_BUSY_CYCLES", 0) <= 0 or not k.startswith("void im::"):
            continue
        cyc = m["SQ_BUSY_CYCLES"] / 32.0
        rows.append((m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), k.replace("void ", "").split("(")[0], len(c["SQ_BUSY_CYCLES"]), cyc,
                     m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024.0), m.get("SQ_WAIT_ANY", 0.0) / max(m.get("SQ_WAVE_CYCLES", 1.0), 1.0),
                     m.get("SQ_INSTS_VALU", 0.0), m.get("SQ_INSTS_LDS", 0.0)))
    with open(os.path.join(dst, f"{tag}_mfma_busy_share.csv"), "w") as fh:
        fh.write("kernel,launches,busy_cycles_per_launch,mfma_pipe_busy_share,wave_cycles_waiting_share,valu_insts_per_launch,lds_insts_per_launch\n")
        for r in sorted(rows, reverse=True):
            fh.write(f'"{r[1]}",{r[2]},{r[3]:.0f},{r[4]:.3f},{r[5]:.2f},{r[6]:.0f},{r[7]:.0f}\n')
for name in ("bench.json", "bench_streams1_under_rocprof.json", "bench_default_under_rocprof.json", "bench_config5.json", "bench_config3.json", "bench_steps20_warmup5.json",
             "bench_config5_under_rocprof.json", "parity_winograd.json", "parity_direct_conv.json", "bench_config4_1gpu_256epochs.json",
             "bench_config4_nccl_world1.json", "bench_2ranks_one_device_gloo.json", "bench_config4_8ranks_one_device_gloo.json",
             "match_call_phases.txt", "sinkhorn.txt", "nms.txt", "assign.txt", "parity_adaptive_10epochs.json", "attn_batch.txt"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        open(os.path.join(dst, f"{tag}_{name}"), "w").write(open(p).read())
print(open(os.path.join(dst, f"{tag}_bench.json")).read()[:3000] if os.path.exists(os.path.join(dst, f"{tag}_bench.json")) else "no bench.json")
