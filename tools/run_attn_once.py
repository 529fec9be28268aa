import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icepy4d_amd import _lib
from icepy4d_amd._lib import ptr, stream_ptr
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = _lib.Context(0)
q = torch.randn(2, 4, n, 64, device="cuda"); k = torch.randn_like(q); v = torch.randn_like(q)
out = torch.empty(2, n, 256, device="cuda"); dn = torch.tensor([n, n], dtype=torch.int32, device="cuda")
for _ in range(5):
    ctx.call("im_flash_attn", ptr(q), ptr(k), ptr(v), ptr(out), ptr(dn), n, 2, 4, 0, 0.125, stream_ptr())
torch.cuda.synchronize()
