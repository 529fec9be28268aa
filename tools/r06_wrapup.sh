#!/bin/bash
# round 6: the launch-mode sweep (SWEEP=1) and the kernel trace of the reference driver's production call
cd "$(dirname "$0")/.."
O=gpurun_out/r06b; mkdir -p $O
export TMPDIR=/tmp
if [ "$SWEEP" = "1" ]; then bash tools/sweep_launch_mode.sh > $O/sweep_launch_mode.txt 2>&1; cat $O/sweep_launch_mode.txt; fi
python3 tools/run_production_call.py > $O/production_call_plain.json 2> $O/production_call_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prodcall -- python3 tools/run_production_call.py > $O/production_call_under_rocprof.json 2> $O/production_call_under_rocprof.err
python3 tools/summarize_production_call.py $O/prodcall $O/production_call_under_rocprof.json $O/production_call_kernel_stats.csv > $O/summary.log 2>&1
rm -rf $O/prodcall
tail -1 $O/production_call_plain.json; head -50 $O/production_call_kernel_stats.csv; tail -5 $O/summary.log
