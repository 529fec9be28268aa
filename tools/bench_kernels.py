#!/usr/bin/env python3
"""Micro-benchmarks of the MFMA kernels at the benchmark shapes (GPU box). Usage: python tools/bench_kernels.py [attn|conv|gemm]..."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icepy4d_amd import _lib  # noqa: E402
from icepy4d_amd._lib import ptr, stream_ptr  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    which = sys.argv[1:] or ["attn", "conv", "gemm"]
    ctx = _lib.Context(0)
    if "attn" in which:
        for n in (4096, 2048, 1024):
            q = torch.randn(2, 4, n, 64, device="cuda"); k = torch.randn_like(q); v = torch.randn_like(q)
            out = torch.empty(2, n, 256, device="cuda")
            dn = torch.tensor([n, n], dtype=torch.int32, device="cuda")
            for cross in (0, 1):
                ms = timeit(lambda: ctx.call("im_flash_attn", ptr(q), ptr(k), ptr(v), ptr(out), ptr(dn), n, 2, 4, cross, 0.125, stream_ptr()))
                fl = 2 * 4 * 4.0 * n * n * 64
                print(f"attn n={n} cross={cross}: {ms:.4f} ms  {fl / ms / 1e9:.1f} TFLOP/s (executed)  {fl / ms / 1e9 / 157.3 * 100:.1f}% of fp32 MFMA peak", flush=True)
    if "attn" in which:   # pruned pairs: the buffers are sized for 4096 keypoints, fewer are live (decided on the device)
        nmax = 4096
        q = torch.randn(2, 4, nmax, 64, device="cuda"); k = torch.randn_like(q); v = torch.randn_like(q)
        out = torch.empty(2, nmax, 256, device="cuda")
        for n in (3000, 2048, 1500, 1000, 500):
            dn = torch.tensor([n, n], dtype=torch.int32, device="cuda")
            ms = timeit(lambda: ctx.call("im_flash_attn", ptr(q), ptr(k), ptr(v), ptr(out), ptr(dn), nmax, 2, 4, 1, 0.125, stream_ptr()))
            fl = 2 * 4 * 4.0 * n * n * 64
            print(f"attn n_max=4096 live n={n} cross=1: {ms:.4f} ms  {fl / ms / 1e9:.1f} TFLOP/s (executed)", flush=True)
    if "conv" in which:
        import ctypes, json
        for (h, w, cin, cout, pool) in ((1080, 1920, 64, 64, 1), (540, 960, 64, 64, 0), (540, 960, 64, 64, 1), (270, 480, 64, 128, 0),
                                        (270, 480, 128, 128, 1), (135, 240, 128, 128, 0), (135, 240, 128, 256, 0)):
            x = torch.randn(2, h, w, cin, device="cuda")
            wt = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
            b = torch.randn(cout)
            ho, wo = (h // 2, w // 2) if pool else (h, w)
            out = torch.empty(2, ho, wo, cout, device="cuda")
            fl = 2 * 2.0 * 9 * cin * cout * h * w
            line = f"conv {h}x{w} {cin}->{cout} pool={pool}:"
            for name in ("im_conv3x3", "im_conv3x3_winograd"):
                ctx.call(name, ptr(x), ptr(wt), ptr(b), ptr(out), 2, h, w, cin, cout, 1, pool, stream_ptr())  # warm-up
                ctx.call("im_profile_begin")
                for _ in range(3):
                    ctx.call(name, ptr(x), ptr(wt), ptr(b), ptr(out), 2, h, w, cin, cout, 1, pool, stream_ptr())
                buf = ctypes.create_string_buffer(4096)
                ctx.call("im_profile_end", buf, len(buf))
                prof = json.loads(buf.value.decode())
                cal = prof.pop("_empty_event_pair", None)
                v = next(iter(prof.values()))
                ms = v["total_ms"] / v["count"] - (cal["total_ms"] / cal["count"] if cal else 0.0)
                line += f"  {name[3:]} {ms:.3f} ms ({fl / ms / 1e9:.0f} TFLOP/s alg.)"
            print(line, flush=True)
    if "gemm" in which:
        for (m, n, k, big) in ((16384, 768, 256, 0), (16384, 768, 256, 1), (16384, 512, 256, 0), (16384, 512, 256, 1), (8192, 256, 256, 0), (8192, 256, 256, 1), (8192, 512, 512, 0), (8192, 512, 512, 1), (8192, 256, 512, 0), (8192, 256, 512, 1), (8192, 768, 256, 0), (8192, 768, 256, 1), (4096, 4096, 256, 1), (64800, 256, 256, 0), (64800, 256, 256, 1)):
            a = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda"); b = torch.randn(n, device="cuda")
            c = torch.empty(m, n, device="cuda")
            ms = timeit(lambda: ctx.call("im_gemm_nt", ptr(a), ptr(w), ptr(b), ptr(c), m, n, k, 1.0, big, stream_ptr()))
            fl = 2.0 * m * n * k
            print(f"gemm {m}x{n}x{k} big={big}: {ms:.4f} ms  {fl / ms / 1e9:.1f} TFLOP/s  {fl / ms / 1e9 / 157.3 * 100:.1f}%", flush=True)


if __name__ == "__main__":
    main()
