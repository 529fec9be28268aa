#!/usr/bin/env python3
"""Attention stage entry at n = 4096 with 2 .. 20 images per launch: does a second resident block per CU pay? (GPU box)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icepy4d_amd import _lib
from icepy4d_amd._lib import ptr, stream_ptr
ctx = _lib.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
zeros = len(sys.argv) > 2 and sys.argv[2] == "zeros"     # all-zero operands: the clock the chip holds without data toggling (guide: DVFS give-back)
for batch in (2, 4, 8, 20):
    q = torch.randn(batch, 4, n, 64, device="cuda"); k = torch.randn_like(q); v = torch.randn_like(q)
    if zeros:
        q.zero_(); k.zero_(); v.zero_()
    out = torch.empty(batch, n, 256, device="cuda"); dn = torch.full((batch,), n, dtype=torch.int32, device="cuda")
    for cross in (0, 1, 2):       # 2: self attention with the f32-input MFMA kernel of rounds 1-5
        f = lambda: ctx.call("im_flash_attn", ptr(q), ptr(k), ptr(v), ptr(out), ptr(dn), n, batch, 4, cross, 0.125, stream_ptr())
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        fl = batch * 4 * 4.0 * n * n * 64
        print(f"n={n} batch={batch} {'zeros ' if zeros else ''}cross={cross & 1} {'f32-input MFMA form' if cross & 2 else 'bf16 planes'}: {ms:.4f} ms = {ms / batch * 2 * 1e3:.1f} us per pair, {fl / ms / 1e9:.1f} TFLOP/s fp32-equivalent", flush=True)
