import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from icepy4d_amd import _lib
from icepy4d_amd._lib import ptr, stream_ptr
import torch.nn.functional as F
ctx = _lib.Context(0)
cin, cout, h, w = 16, 64, 8, 32
g = torch.Generator().manual_seed(1)
x = torch.randn(1, cin, h, w, generator=g)
wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
b = torch.zeros(cout)
ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
dx = x.permute(0, 2, 3, 1).contiguous().cuda()
out = torch.full((1, h, w, cout), float("nan"), device="cuda")
ctx.call("im_conv3x3_winograd", ptr(dx), ptr(wt.contiguous()), ptr(b), ptr(out), 1, h, w, cin, cout, 0, 0, stream_ptr())
torch.cuda.synchronize()
o = out.cpu().permute(0, 3, 1, 2).double()
err = (o - ref).abs()
print("max err", err.max().item())
# contribution split: conv with only channels 0..7 (slab 0) / 8..15 (slab 1)
r0 = F.conv2d(x[:, :8].double(), wt[:, :8].double(), padding=1)
r1 = F.conv2d(x[:, 8:].double(), wt[:, 8:].double(), padding=1)
for name, cand in (("slab0 only", r0), ("slab1 only", r1), ("2*slab0", 2 * r0), ("2*slab1", 2 * r1), ("slab0(x)+slab1 with U swapped", None)):
    if cand is not None:
        print(name, (o - cand).abs().max().item())
x0u1 = F.conv2d(x[:, :8].double(), wt[:, 8:].double(), padding=1)
x1u0 = F.conv2d(x[:, 8:].double(), wt[:, :8].double(), padding=1)
print("x0*U1 + x1*U0", (o - (x0u1 + x1u0)).abs().max().item())
print("x0*U0 + x1*U0", (o - (r0 + x1u0)).abs().max().item(), " x0*U1 + x1*U1", (o - (x0u1 + r1)).abs().max().item())
print("x0*U0 + x0*U1", (o - (r0 + x0u1)).abs().max().item(), " x1*U0 + x1*U1", (o - (x1u0 + r1)).abs().max().item())
print("err by out channel half:", err[0, :32].max().item(), err[0, 32:].max().item(), " by rows:", [round(err[0, :, y].max().item(), 3) for y in range(h)])
