#!/bin/bash
# N consecutive passes of the GPU suite in fresh processes, with the guard words of IM_DEBUG_GUARDS=1 around every device buffer
# of the library (GPU AddressSanitizer / XNACK are not available on this pool: this is the substitute), each pass under
# `python -X faulthandler`; the full log of a pass is kept only when it fails. Writes gpurun_out/<tag>/tally.json.
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/stress_gpu_suite.sh 10 1 stressA'
# Arguments: passes (default 20), guards 1|0 (default 1), tag (default stress).
n=${1:-20}
guards=${2:-1}
tag=${3:-stress}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
echo "[" > $out/tally.json
fails=0
for i in $(seq 1 $n); do
  t0=$(date +%s)
  IM_DEBUG_GUARDS=$guards timeout 1200 python3 -X faulthandler -m pytest tests -m gpu -x -q -p no:cacheprovider > $out/pass_$i.log 2>&1
  rc=$?
  t1=$(date +%s)
  summary=$(grep -E "passed|failed|error" $out/pass_$i.log | tail -1 | tr -d '"')
  gf=$(grep -o "IM_DEBUG_GUARDS: [0-9]* guard failure" $out/pass_$i.log | tail -1 | grep -o "[0-9]*" | head -1)
  [ -n "$i" ] && [ $i -gt 1 ] && echo "," >> $out/tally.json
  echo "{\"pass\": $i, \"rc\": $rc, \"seconds\": $((t1 - t0)), \"guards\": $guards, \"guard_failures\": ${gf:-null}, \"summary\": \"$summary\"}" >> $out/tally.json
  echo "pass $i rc $rc $((t1 - t0)) s guard_failures ${gf:-n/a}: $summary"
  if [ $rc -ne 0 ]; then
    fails=$((fails + 1))
    tail -60 $out/pass_$i.log > $out/FAILED_pass_$i.tail
  else
    tail -4 $out/pass_$i.log > $out/pass_$i.tail
    rm -f $out/pass_$i.log
  fi
done
echo "]" >> $out/tally.json
echo "passes $n failed $fails"
