#!/usr/bin/env python3
"""Sinkhorn solve (`im_log_optimal_transport`, 20 iterations) at n x n couplings for the kernel paths the library
can be switched to (environment read once per process, so every setting runs in a child process). Times the stage entry point with
HIP events over `reps` solves (the final materialisation of the (n+1)^2 output included: one more read + write of the matrix, so the
per-iteration figure is (t_20 - t_0) / 20 from a second run with 0 iterations), and checks every form against the first one.

    python tools/bench_sinkhorn.py [n=16384] [reps=5]
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETTINGS = [("default (one exponential per element, repair folded into the combine kernel: 2 launches per iteration)", {}),
            ("two sweeps (round 1; the fallback for n > 16384 / unaligned rows)", {"IM_SINKHORN_TWO_SWEEP": "1"})]


def child(n, reps):
    sys.path.insert(0, ROOT)
    import torch
    from icepy4d_amd._lib import ptr
    from icepy4d_amd.engine import Engine
    e = Engine(0)
    e.reserve(64, 64, 2, n)
    g = torch.Generator(device="cuda").manual_seed(7)
    z = torch.randn(n, n, device="cuda", generator=g) * 2
    out = torch.empty((n + 1) * (n + 1), device="cuda")
    res = {}
    for iters in (20, 0):
        for _ in range(2):
            e.ctx.call("im_log_optimal_transport", ptr(z), n, n, n, 1.0, iters, ptr(out), e.stream_ptr())
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(reps):
            e.ctx.call("im_log_optimal_transport", ptr(z), n, n, n, 1.0, iters, ptr(out), e.stream_ptr())
        t1.record()
        torch.cuda.synchronize()
        res[iters] = t0.elapsed_time(t1) / reps
        if iters == 20:
            o = out.view(n + 1, n + 1)
            res["sample"] = o[::997, ::991].double().cpu().numpy().round(5).tolist()
            res["col_lse_max_abs"] = float(torch.logsumexp(o[:, :n].double(), 0).abs().max())
    print(json.dumps(res))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    ref = None
    for name, env in SETTINGS:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n), str(reps)], env=dict(os.environ, **env),
                           capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(f"{name:62s} FAILED {r.stderr[-400:]}")
            continue
        d = json.loads(line[-1])
        import numpy as np
        s = np.array(d["sample"])
        if ref is None:
            ref = s
        per_it = (d["20"] - d["0"]) / 20.0
        gb = (n + 1) * (n + 1) * 4 / 1e9
        print(f"{name:62s} solve {d['20']:7.3f} ms, per iteration {per_it * 1e3:7.1f} us = {gb / per_it:6.2f} TB/s of one read; "
              f"max |column lse| {d['col_lse_max_abs']:.1e}; max |diff to first| {np.abs(s - ref).max():.1e}", flush=True)


if __name__ == "__main__":
    if "--child" in sys.argv:
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        main()
