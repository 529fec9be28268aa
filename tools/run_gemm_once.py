#!/usr/bin/env python3
"""A few launches of im_gemm_nt at one shape, for rocprofv3 --pmc passes. Usage: python tools/run_gemm_once.py [m n k big]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icepy4d_amd import _lib  # noqa: E402
from icepy4d_amd._lib import ptr, stream_ptr  # noqa: E402

m, n, k, big = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (64800, 256, 256, 0)))
ctx = _lib.Context(0)
a = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda"); b = torch.randn(n, device="cuda")
c = torch.empty(m, n, device="cuda")
for _ in range(200):   # settle clocks
    ctx.call("im_gemm_nt", ptr(a), ptr(w), ptr(b), ptr(c), m, n, k, 1.0, big, stream_ptr())
torch.cuda.synchronize()
