#!/usr/bin/env python3
"""The assignment stage of LightGlue (`im_assign_from_sim`: lse_stats, col_lse_combine, best_sweep, col_best_combine, filter_scatter) on a
4096 x 4096 similarity matrix, ONE pair per launch: time per call by HIP events, and a SHA-1 of the outputs (to compare builds bit for bit).
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel durations.

    python tools/bench_assign.py [n=4096] [reps=50]
"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icepy4d_amd._lib import ptr  # noqa: E402
from icepy4d_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
e = Engine(0)
e.reserve(64, 64, 2, n)
g = torch.Generator(device="cuda").manual_seed(5)
sim = torch.randn(n, n, device="cuda", generator=g) * 4
z0 = torch.randn(n, device="cuda", generator=g) * 2 + 2
z1 = torch.randn(n, device="cuda", generator=g) * 2 + 2
m0 = torch.zeros(n, dtype=torch.int32, device="cuda"); m1 = torch.zeros_like(m0)
s0 = torch.zeros(n, device="cuda"); s1 = torch.zeros_like(s0)


def call():
    e.ctx.call("im_assign_from_sim", ptr(sim), n, n, n, ptr(z0), ptr(z1), 0.1, ptr(m0), ptr(m1), ptr(s0), ptr(s1), e.stream_ptr())


for _ in range(5):
    call()
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(reps):
    call()
t1.record()
torch.cuda.synchronize()
sha = hashlib.sha1(b"".join(t.cpu().numpy().tobytes() for t in (m0, m1, s0, s1))).hexdigest()[:16]
print(f"im_assign_from_sim {n} x {n}: {1e3 * t0.elapsed_time(t1) / reps:.1f} us per call, {int((m0 > -1).sum())} matches, outputs sha1 {sha}")
