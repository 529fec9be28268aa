#!/usr/bin/env python3
"""In-kernel timeline of conv_wino_bx2.hip from a DIAGNOSTIC build (tools/build_conv_variant.sh y_STAMP -DIM_YSTAMP): shader-clock stamps of wave 0
of every (persistent) block at conv2a's shape (2 x 540 x 960 x 64 -> 64), median over blocks of every interval, in cycles.
    ICEMATCH_LIB=build_abl/y_STAMP/libicematch.so IM_CONV_BX2=1 python tools/conv_bx2_stamps.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icepy4d_amd import _lib
from icepy4d_amd._lib import ptr, stream_ptr
ctx = _lib.Context(0)
cin = int(os.environ.get("CIN", 64)); cout = int(os.environ.get("COUT", 64)); h, w = 540, 960
x = torch.randn(2, h, w, cin, device="cuda"); wt = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5; b = torch.randn(cout)
out = torch.empty(2, h, w, cout, device="cuda")
for _ in range(3):
    ctx.call("im_conv3x3_winograd", ptr(x), ptr(wt), ptr(b), ptr(out), 2, h, w, cin, cout, 1, 0, stream_ptr())
torch.cuda.synchronize()
buf = np.zeros(256 * 256, dtype=np.uint64)
ctx.lib.im_debug_ystamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert ctx.lib.im_debug_ystamps(buf.ctypes.data, buf.size) == 0
st = buf.reshape(256, 256).astype(np.int64)
nq = 4 * (cin // 16)
per_item = 3 * nq + 2
med = lambda d: int(np.median(d))
print(f"prologue (transfers, head, barriers): {med(st[:, 1] - st[:, 0])} cycles")
prev = st[:, 1]
i = 2
for item in range(min(4, (256 - 2) // per_item)):
    steps = vm = bar = 0
    detail = []
    for t in range(nq):
        a, b_, c = st[:, i], st[:, i + 1], st[:, i + 2]
        steps += med(a - prev); vm += med(b_ - a); bar += med(c - b_)
        detail.append(f"{med(a - prev)}/{med(b_ - a)}/{med(c - b_)}")
        prev = c; i += 3
    e0, e1 = st[:, i], st[:, i + 1]
    print(f"item {item}: steps {steps}  vmcnt waits {vm}  barriers {bar}  epilogue {med(e1 - e0)}  total {steps + vm + bar + med(e1 - e0)}")
    print("   quarters steps/vmcnt/barrier: " + "  ".join(detail))
    prev = e1; i += 2
