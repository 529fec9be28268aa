#!/bin/bash
# round 6 dev loop: the Winograd layer tests on both forms, then per-kernel SuperPoint times (BX vs IM_CONV_F32=1)
cd "$(dirname "$0")/.."
O=gpurun_out/conv_bx_dev
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv3x3" > $O/pytest_conv.log 2>&1
tail -5 $O/pytest_conv.log
timeout 600 python tools/time_superpoint_kernels.py > $O/tsk_bx.txt 2>&1
IM_CONV_F32=1 timeout 600 python tools/time_superpoint_kernels.py > $O/tsk_f32.txt 2>&1
echo "== BX"; cat $O/tsk_bx.txt
echo "== F32"; cat $O/tsk_f32.txt
