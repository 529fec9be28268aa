#!/usr/bin/env python3
"""In-kernel timeline of gemm.hip's proj_rows_kernel from a DIAGNOSTIC build (tools/build_variant_file.sh p_STAMP gemm.hip -DIM_PSTAMP): shader-clock stamps
of wave 0 of every block of ONE LightGlue self block's qkv projection at PAIRS pairs of 4096 keypoints per launch, median over blocks, in cycles.
    ICEMATCH_LIB=build_abl/p_STAMP/libicematch.so IM_PROJ_TILED=0 PAIRS=10 python tools/proj_stamps.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icepy4d_amd import synthetic
from icepy4d_amd.engine import Engine
pairs = int(os.environ.get("PAIRS", 10)); K = 4096
e = Engine(0)
e.load_state_dict("lightglue", synthetic.lightglue_state_dict(0, "passthrough"))
e.reserve(64, 64, 2 * pairs, K)
g = torch.Generator().manual_seed(0)
e.kpts[:2 * pairs] = (torch.rand(2 * pairs, K, 2, generator=g) * 1000).cuda()
d = torch.randn(2 * pairs, K, 256, generator=g); d = d / d.norm(dim=-1, keepdim=True)
e.desc[:2 * pairs] = d.cuda(); e.n[:2 * pairs] = K
for _ in range(2):
    e.lightglue((1920.0, 1080.0), (1920.0, 1080.0), n_pairs=pairs)
torch.cuda.synchronize()
buf = np.zeros(4096 * 16, dtype=np.uint64)
e.ctx.lib.im_debug_pstamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert e.ctx.lib.im_debug_pstamps(buf.ctypes.data, buf.size) == 0
nb = min(4096, (K // 32) * 2 * pairs)
st = buf.reshape(4096, 16).astype(np.int64)[:nb]
st = st[st[:, 0] > 0]
names = ["stage x (load, cut, planes, barrier)", "tile 0: 16 steps", "tile 0: epilogue", "tile 1: 16 steps", "tile 1: epilogue", "tile 2: 16 steps", "tile 2: epilogue"]
ntile = 3 if (st[:, 6] > st[:, 5]).all() else 2       # the last launch that wrote the stamps: the final layer's cross projection has two tiles
print(f"{len(st)} blocks stamped, {pairs} pairs per launch, last projection launch of the forward ({'qkv' if ntile == 3 else 'cross'}: {ntile} tiles per wave)")
tot = 0
for i, n in enumerate(names[:1 + 2 * ntile]):
    dd = st[:, i + 1] - st[:, i]
    print(f"  {n:40s} median {int(np.median(dd)):7d}  p10 {int(np.percentile(dd, 10)):7d}  p90 {int(np.percentile(dd, 90)):7d}")
    tot += int(np.median(dd))
print(f"  sum of medians {tot}")
