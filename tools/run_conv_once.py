import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icepy4d_amd import _lib
from icepy4d_amd._lib import ptr, stream_ptr
name = sys.argv[1] if len(sys.argv) > 1 else "im_conv3x3_winograd"
h, w, cin, cout, pool = 1080, 1920, 64, 64, 1
ctx = _lib.Context(0)
x = torch.randn(2, h, w, cin, device="cuda"); wt = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5; b = torch.randn(cout)
out = torch.empty(2, h // 2, w // 2, cout, device="cuda")
for _ in range(3):
    ctx.call(name, ptr(x), ptr(wt), ptr(b), ptr(out), 2, h, w, cin, cout, 1, pool, stream_ptr())
torch.cuda.synchronize()
