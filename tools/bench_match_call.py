#!/usr/bin/env python3
"""Wall time of the drop-in call `matcher.match(image0, image1, ...)` (host arrays in, numpy results out), the way
icepy4d's `main_dev.py:115-132` uses it. Usage: python tools/bench_match_call.py [H W K reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icepy4d_amd import matching, synthetic

H, W, K, REPS = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (1080, 1920, 4096, 5)))
a, b = synthetic.translated_pair(0, H, W, 16, 8)
m = matching.LightGlueMatcher({"state_dicts": {"superpoint": synthetic.superpoint_state_dict(0),
                                               "lightglue": synthetic.lightglue_state_dict(0, "passthrough")}})
cases = (("NONE (whole image, HIP-graph replay)", matching.TileSelection.NONE, matching.GeometricVerification.NONE),
         ("GRID 1x1 (tile path)", matching.TileSelection.GRID, matching.GeometricVerification.NONE),
         ("GRID 1x1 + device RANSAC", matching.TileSelection.GRID, matching.GeometricVerification.PYDEGENSAC))
for name, sel, gv in cases:
    for r in range(REPS):
        t0 = time.perf_counter()
        m.match(a, b, quality=matching.Quality.HIGH, tile_selection=sel, grid=[1, 1], overlap=0,
                max_keypoints=K, geometric_verification=gv, threshold=2)
        dt = time.perf_counter() - t0
        print(f"{name:38s} rep {r}: {dt * 1e3:.1f} ms, {len(m.mkpts0)} points", flush=True)

# the reference's own pattern: a fresh matcher object per epoch (`main_dev.py:115-132`)
sds = {"superpoint": synthetic.superpoint_state_dict(0), "lightglue": synthetic.lightglue_state_dict(0, "passthrough")}
for r in range(4):
    t0 = time.perf_counter()
    m2 = matching.LightGlueMatcher({"state_dicts": sds})
    t1 = time.perf_counter()
    m2.match(a, b, quality=matching.Quality.HIGH, tile_selection=matching.TileSelection.NONE, max_keypoints=K,
             geometric_verification=matching.GeometricVerification.NONE)
    print(f"fresh matcher per call                 rep {r}: construct {(t1 - t0) * 1e3:.2f} ms + match {(time.perf_counter() - t1) * 1e3:.1f} ms", flush=True)

if os.environ.get("IM_BENCH_TILES", "1") == "1":
    # production-like call (`main_dev.py:115-132`): 12 MP RGB pair, 3 x 3 grid with overlap, 8192 keypoints per tile
    a, b = synthetic.translated_pair(1, 3000, 4000, 48, 16)
    a3, b3 = np.repeat(a[:, :, None], 3, 2), np.repeat(b[:, :, None], 3, 2)
    for name, sel in (("12 MP RGB, GRID 3x3, 8192 kpts/tile", matching.TileSelection.GRID),
                      ("12 MP RGB, PRESELECTION 3x3, 8192 kpts/tile", matching.TileSelection.PRESELECTION)):
        for r in range(3):
            t0 = time.perf_counter()
            m.match(a3, b3, quality=matching.Quality.HIGH, tile_selection=sel, grid=[3, 3], overlap=200,
                    max_keypoints=8192, geometric_verification=matching.GeometricVerification.NONE)
            dt = time.perf_counter() - t0
            print(f"{name:46s} rep {r}: {dt * 1e3:.1f} ms, {len(m.mkpts0)} points", flush=True)

if os.environ.get("IM_BENCH_PRODUCTION", "1") == "1":
    # the call of `main_dev.py:115-132` with its own parameters, on a synthetic 24 MP RGB pair whose texture survives the two
    # pyramid levels of the preselection pass (a translated 1000 x 1500 pair, every pixel blown up to a 4 x 4 block)
    ha, hb = synthetic.translated_pair(3, 1000, 1500, 24, 8, noise=0.0)
    a3 = np.repeat(np.kron(ha, np.ones((4, 4), np.uint8))[:, :, None], 3, 2)
    b3 = np.repeat(np.kron(hb, np.ones((4, 4), np.uint8))[:, :, None], 3, 2)
    for r in range(3):
        t0 = time.perf_counter()
        m.match(a3, b3, quality=matching.Quality.HIGH, tile_selection=matching.TileSelection.PRESELECTION, grid=[2, 2], overlap=200,
                origin=[0, 0], min_matches_per_tile=3, max_keypoints=8196,
                geometric_verification=matching.GeometricVerification.PYDEGENSAC, threshold=2, confidence=0.9999)
        dt = time.perf_counter() - t0
        print(f"main_dev.py call, 24 MP RGB, PRESELECTION 2x2   rep {r}: {dt * 1e3:.1f} ms, {len(m.mkpts0)} points; "
              + ", ".join(f"{k} {v * 1e3:.1f}" for k, v in m.timer.times.items()), flush=True)
