#!/bin/bash
# PMC passes + kernel trace of the attention stage entry (tools/run_attn_once.py); run on the GPU box from the repo root:
#   bash tools/prof_attn_bx.sh gpurun_out/attn_bx_pmc
root=$PWD; out=$root/${1:-gpurun_out/attn_bx_pmc}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/tools/run_attn_once.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out/pmc1 -- python3 $root/tools/run_attn_once.py > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SALU --kernel-trace --output-format csv -d $out/pmc2 -- python3 $root/tools/run_attn_once.py > /dev/null 2>&1
cd $root
for d in pmc1 pmc2; do f=$(find $out/$d -name "*counter_collection.csv" | head -1); echo "== $d"; python3 tools/pmc_parse.py $f flash_attn kv_planes; done
f=$(find $out/trace -name "*kernel_stats.csv" | head -1); head -5 $f
