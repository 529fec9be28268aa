#!/bin/bash
# round 6 dev loop for conv_wino_bx2.hip: the Winograd layer tests with IM_CONV_BX2=1, bit-identity against BX, per-kernel SuperPoint times
cd "$(dirname "$0")/.."
O=gpurun_out/conv_bx2_dev
mkdir -p $O
IM_CONV_BX2=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv3x3_winograd and bf16x6" > $O/pytest_conv.log 2>&1
tail -5 $O/pytest_conv.log
timeout 300 python tools/conv_bx2_identity.py 2>&1 | tail -12
IM_CONV_BX2=1 timeout 600 python tools/time_superpoint_kernels.py "$@" > $O/tsk_bx2.txt 2>&1
timeout 600 python tools/time_superpoint_kernels.py > $O/tsk_bx.txt 2>&1
echo "== BX2"; cat $O/tsk_bx2.txt
echo "== BX"; cat $O/tsk_bx.txt
