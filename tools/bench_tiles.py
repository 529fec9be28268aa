#!/usr/bin/env python3
"""Tile-mode `match()` (the production call of icepy4d, `main_dev.py:115-132`) on one synthetic pair: wall time per call and
the matcher's own timer breakdown. Usage: python tools/bench_tiles.py [H W grid overlap max_keypoints reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icepy4d_amd import matching, synthetic

H, W, G, OV, K, REPS = (int(x) for x in (sys.argv[1:7] if len(sys.argv) >= 7 else (3000, 4000, 3, 200, 8192, 3)))
a, b = synthetic.translated_pair(0, H, W, 16, 8)
sds = {"superpoint": synthetic.superpoint_state_dict(0), "lightglue": synthetic.lightglue_state_dict(0, "passthrough")}
for host in (True, False):
  m = matching.LightGlueMatcher({"state_dicts": sds, "host_tile_merge": host})
  print("host-side merge (per-pair round trips)" if host else "device-side merge (one selection)")
  for sel in (matching.TileSelection.GRID,):
    for r in range(REPS):
        t0 = time.perf_counter()
        m.match(a, b, quality=matching.Quality.HIGH, tile_selection=sel, grid=[G, G], overlap=OV, max_keypoints=K,
                geometric_verification=matching.GeometricVerification.NONE)
        dt = time.perf_counter() - t0
        print(f"{sel.name} rep {r}: {dt * 1e3:.1f} ms, {len(m.mkpts0)} matches", flush=True)
