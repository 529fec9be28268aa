#!/bin/bash
# Runs the measurement passes behind profiles/ on the GPU box (one gpurun call):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01'
# rocprofv3 wants cwd=/tmp and TMPDIR=/tmp on this pool; PMC passes are separate runs without other trace domains.
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $root/bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1 -- python3 $root/bench.py --no-cpu-baseline --streams 1 > $out/bench_streams1_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats3 -- python3 $root/bench.py --no-cpu-baseline > $out/bench_default_under_rocprof.json 2> /dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out/pmc_attn_sq -- python3 $root/tools/run_attn_once.py > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_attn_fetch -- python3 $root/tools/run_attn_once.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_attn_write -- python3 $root/tools/run_attn_once.py > /dev/null 2>&1
if [ -x $root/build_abl/mfma_peak ]; then timeout 200 $root/build_abl/mfma_peak > $out/mfma_peak.txt 2>&1; fi
cd $root
python3 - <<PY
import csv, glob, collections, json
out = "$out"
for name in ("pmc_attn_sq", "pmc_attn_fetch", "pmc_attn_write"):
    agg = collections.defaultdict(list)
    for fn in glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if "flash_attn" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(name, {k: (sum(v) / len(v), len(v)) for k, v in agg.items()})
for name in ("stats1", "stats3"):
    for fn in glob.glob(f"{out}/{name}/**/*kernel_stats.csv", recursive=True):
        rows = list(csv.DictReader(open(fn)))
        print(name, [(r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"]) for r in rows[:6]])
print(open(f"{out}/bench.json").read()[-2500:])
PY
