#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (imports oracle/): geometric verification (row f-2) with the device hypothesis stage vs the oracle's numpy
RANSAC in its place, on synthetic two-view matches. Run from the repo root on the GPU box: python tests/bench_gv.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icepy4d_amd.engine import Engine
from icepy4d_amd.matching import GeometricVerification, geometric_verification
from oracle import gv_cpu

e = Engine(0)
for n_pts, frac in ((1000, 0.2), (4000, 0.5), (10000, 0.5)):
    rng = np.random.default_rng(0)
    n_out = int(n_pts * frac)
    X = np.c_[rng.uniform(-1, 1, n_pts), rng.uniform(-1, 1, n_pts), rng.uniform(4, 8, n_pts)]
    Kc = np.array([[800, 0, 320], [0, 800, 240], [0, 0, 1.0]])
    p0 = (Kc @ X.T).T; p0 = p0[:, :2] / p0[:, 2:]
    p1 = (Kc @ (X + np.array([0.5, 0.05, 0.1])).T).T; p1 = p1[:, :2] / p1[:, 2:] + rng.normal(0, 0.05, size=(n_pts, 2))
    p1[:n_out] += rng.uniform(20, 60, size=(n_out, 2))
    p0, p1 = p0.astype(np.float32), p1.astype(np.float32)
    geometric_verification(p0, p1, GeometricVerification.PYDEGENSAC, threshold=1.0, engine=e)  # warm
    t0 = time.perf_counter(); _, md = geometric_verification(p0, p1, GeometricVerification.PYDEGENSAC, threshold=1.0, engine=e); td = time.perf_counter() - t0
    t0 = time.perf_counter(); _, mh = geometric_verification(p0, p1, GeometricVerification.PYDEGENSAC, threshold=1.0, hypothesis_fn=gv_cpu.hypothesis_fn(p0, p1, 1.0)); th = time.perf_counter() - t0
    print(f"S={n_pts} outliers={frac:.0%}: device {td * 1e3:.2f} ms ({int(md.sum())} inliers), numpy {th * 1e3:.1f} ms ({int(mh.sum())} inliers)")
