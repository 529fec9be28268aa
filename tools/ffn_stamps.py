#!/usr/bin/env python3
"""In-kernel timeline of ffn_fused.hip from a DIAGNOSTIC build (tools/build_ffn_variant.sh f_STAMP -DIM_FSTAMP): shader-clock stamps of wave 0 of every
block, median over blocks of every interval, in cycles; PAIRS pairs of 4096 keypoints per launch.
    ICEMATCH_LIB=build_abl/f_STAMP/libicematch.so PAIRS=2 python tools/ffn_stamps.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icepy4d_amd import _lib
from icepy4d_amd._lib import ptr, stream_ptr
ctx = _lib.Context(0)
pairs = int(os.environ.get("PAIRS", 2)); rows = int(os.environ.get("ROWS", 4096)); nimg = 2 * pairs
g = torch.Generator().manual_seed(1)
x = torch.randn(nimg, rows, 256, generator=g).cuda(); att = torch.randn(nimg, rows, 256, generator=g).cuda()
w0 = (torch.randn(512, 512, generator=g) / 512 ** 0.5).numpy(); b0 = (torch.randn(512, generator=g) * 0.1).numpy()
lg = (1 + 0.1 * torch.randn(512, generator=g)).numpy(); lb = (0.1 * torch.randn(512, generator=g)).numpy()
w3 = (torch.randn(256, 512, generator=g) / 512 ** 0.5).numpy(); b3 = (torch.randn(256, generator=g) * 0.1).numpy()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(3):
    ctx.call("im_ffn_fused", 0, ptr(x), ptr(att), ptr(w0), ptr(b0), ptr(lg), ptr(lb), ptr(w3), ptr(b3), nimg, rows, None, stream_ptr())
torch.cuda.synchronize()
buf = np.zeros(4096 * 16, dtype=np.uint64)
ctx.lib.im_debug_fstamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert ctx.lib.im_debug_fstamps(buf.ctypes.data, buf.size) == 0
st = buf.reshape(4096, 16).astype(np.int64)
nb = (rows + 31) // 32 * nimg
st = st[:min(nb, 4096)]
env = os.environ.get("IM_FFN_SPLIT")
split = env == "1" if env else nb > torch.cuda.get_device_properties(0).multi_processor_count
names = (["stage x + fetch att", "GEMM1 half 0", "barrier + put att + barrier", "GEMM1 half 1", "barrier + hidden rows + barrier", "LayerNorm + GELU", "barrier + put h0 + barrier",
          "GEMM2 half 0", "barrier + put h1 + resid + barrier", "GEMM2 half 1", "stores"] if split else
         ["stage", "GEMM1", "barrier + hidden rows + barrier", "LayerNorm + GELU", "barrier + put + barrier", "GEMM2", "stores"])
print(f"{'split form (two blocks per CU)' if split else 'full-K form (one block per CU)'}: {nb} blocks of 32 rows, {pairs} pairs per launch")
tot = 0
for i, n in enumerate(names):
    d = st[:, i + 1] - st[:, i]
    print(f"  {n:36s} median {int(np.median(d)):7d}  p10 {int(np.percentile(d, 10)):7d}  p90 {int(np.percentile(d, 90)):7d}")
    tot += int(np.median(d))
print(f"  sum of medians {tot}; block total median {int(np.median(st[:, len(names)] - st[:, 0]))} cycles (s_memtime)")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for it in range(5):
    e0.record(); ctx.call("im_ffn_fused", 0, ptr(x), ptr(att), ptr(w0), ptr(b0), ptr(lg), ptr(lb), ptr(w3), ptr(b3), nimg, rows, None, stream_ptr()); e1.record()
    torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print(f"  the entry (weights packed and uploaded inside: not the kernel alone) {min(ts):.3f} ms")
