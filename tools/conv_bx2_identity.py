#!/usr/bin/env python3
"""conv_wino_bx2.hip computes every output with the arithmetic of conv_wino.hip BX (same V, same cuts, same order of the six products and of the
chunks, same epilogue): the outputs of the two kernels must be EQUAL BIT FOR BIT, at every shape and with pooling. Run on the GPU."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icepy4d_amd import _lib
from icepy4d_amd._lib import ptr, stream_ptr
ctx = _lib.Context(0)
bad = 0
for (cin, cout, h, w, pool) in [(64, 64, 37, 70, 0), (64, 64, 44, 96, 1), (64, 128, 40, 64, 1), (128, 128, 20, 32, 0), (128, 256, 17, 33, 0), (64, 64, 6, 64, 1),
                                (64, 64, 270, 480, 1), (128, 128, 135, 240, 0), (64, 64, 540, 960, 0)]:
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(2, h, w, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    ho, wo = (h // 2, w // 2) if pool else (h, w)
    outs = []
    for form in ("0", "1"):
        os.environ["IM_CONV_BX2"] = form
        out = torch.full((2, ho, wo, cout), float("nan"), device="cuda")
        ctx.call("im_conv3x3_winograd", ptr(x), ptr(wt), ptr(b), ptr(out), 2, h, w, cin, cout, 1, pool, stream_ptr())
        torch.cuda.synchronize()
        outs.append(out)
    same = torch.equal(outs[0], outs[1])
    nd = (outs[0] != outs[1]).sum().item()
    print(f"{cin:4d}->{cout:4d} {h:4d}x{w:4d} pool={pool}: {'identical' if same else f'DIFFERENT in {nd} values, max |d| ' + str((outs[0] - outs[1]).abs().max().item())}")
    bad += not same
sys.exit(1 if bad else 0)
