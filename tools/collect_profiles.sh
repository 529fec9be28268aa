#!/bin/bash
# Runs the measurement passes behind profiles/ on the GPU box (one gpurun call):
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/collect_profiles.sh r05'
# rocprofv3 wants cwd=/tmp and TMPDIR=/tmp on this pool; PMC passes are separate runs with --kernel-trace only.
# A second argument "core" re-measures only what a kernel change moves (bench lines, traces, counters of the LightGlue pair, parity
# report) and leaves the rehearsal / SuperGlue / per-phase files of the last full collection in place.
tag=${1:-r05}
mode=${2:-full}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $root/bench.py > $out/bench.json 2> $out/bench.err
python3 $root/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-side-measurements > $out/bench_steps20_warmup5.json 2> /dev/null   # the driver's flags
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1 -- python3 $root/bench.py --no-cpu-baseline --no-side-measurements --streams 1 --batch 1 > $out/bench_streams1_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats10 -- python3 $root/bench.py --no-cpu-baseline --no-side-measurements --streams 1 --batch 10 > $out/bench_streams1_batch10_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_default -- python3 $root/bench.py --no-cpu-baseline --no-side-measurements > $out/bench_default_under_rocprof.json 2> /dev/null
python3 $root/bench.py --config 5 > $out/bench_config5.json 2> $out/bench_config5.err
python3 $root/bench.py --config 3 --no-cpu-baseline > $out/bench_config3.json 2> $out/bench_config3.err
if [ "$mode" = full ]; then
# config 4 rehearsed on the one GPU of this box: (a) 256 of its 2048 epochs on one rank (98 KB records with keypoints);
# (b) the same through torch.distributed.run with ONE rank and the nccl backend forced (RCCL init, all-gather, `ranks` object);
# (c) two ranks sharing cuda:0 over gloo (IM_BENCH_ONE_DEVICE=1): the N > 1 code path end to end with real GPU work
python3 $root/bench.py --config 4 --steps 256 --no-cpu-baseline --no-side-measurements > $out/bench_config4_1gpu_256epochs.json 2> $out/bench_config4.err
IM_BENCH_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 $root/bench.py --gpus 1 --config 4 --steps 64 --no-cpu-baseline --no-side-measurements > $out/bench_config4_nccl_world1.json 2> $out/bench_config4_nccl.err
IM_BENCH_ONE_DEVICE=1 python3 $root/bench.py --gpus 2 --steps 40 --warmup 10 --no-cpu-baseline --no-side-measurements > $out/bench_2ranks_one_device_gloo.json 2> $out/bench_2ranks.err
# the shape of the first 8-GPU lease on the one device of this box: eight ranks, gloo, 32 pairs each of configs[3]
IM_BENCH_ONE_DEVICE=1 python3 $root/bench.py --gpus 8 --config 4 --steps 32 --warmup 4 --no-cpu-baseline --no-side-measurements > $out/bench_config4_8ranks_one_device_gloo.json 2> $out/bench_8ranks.err
python3 $root/tools/profile_match_call.py > $out/match_call_phases.txt 2>&1
python3 $root/tools/bench_sinkhorn.py > $out/sinkhorn.txt 2>&1
python3 $root/tools/bench_nms.py > $out/nms.txt 2>&1
python3 $root/tools/bench_assign.py > $out/assign.txt 2>&1
python3 $root/tools/bench_assign_pairs.py >> $out/assign.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_assign -- python3 $root/tools/bench_assign.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_config5 -- python3 $root/bench.py --config 5 --steps 3 --warmup 1 > $out/bench_config5_under_rocprof.json 2> /dev/null
fi
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_lg_$c -- python3 $root/tools/run_pair_once.py lightglue 2 > /dev/null 2>&1
  [ "$mode" = full ] && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_sg_$c -- python3 $root/tools/run_pair_once.py superglue 1 > /dev/null 2>&1
done
python3 $root/tools/bench_attn_batch.py 4096 > $out/attn_batch.txt 2>&1          # the attention stage entry, 1 .. 10 pairs per launch, both forms
python3 $root/tools/bench_attn_batch.py 4096 zeros >> $out/attn_batch.txt 2>&1   # the same on all-zero operands: the clock without data toggling
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out/pmc_attn_sq -- python3 $root/tools/run_attn_once.py > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $out/pmc_sq_all -- python3 $root/tools/run_pair_once.py lightglue 2 > /dev/null 2>&1
cd $root
python3 tests/parity_report.py --epochs 30 --superglue --out $out/parity_winograd.json > $out/parity_winograd.log 2>&1
python3 tests/parity_report.py --adaptive 10 --out $out/parity_adaptive_10epochs.json > $out/parity_adaptive.log 2>&1
[ "$mode" = full ] && IM_CONV_DIRECT=1 python3 tests/parity_report.py --out $out/parity_direct_conv.json > $out/parity_direct_conv.log 2>&1
python3 tools/summarize_profiles.py $tag
