#!/usr/bin/env python3
"""Where the milliseconds of one `LightGlueMatcher.match()` call go (1080p pair, 4096 keypoints, host arrays in / numpy out):
the phases of `_match_images` timed one by one on a warmed-up matcher - stacking + upload, graph replay + synchronise, downloads."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from icepy4d_amd import matching, synthetic

H, W, K = 1080, 1920, 4096
a, b = synthetic.stereo_pair(0, H, W)
m = matching.LightGlueMatcher({"state_dicts": {"superpoint": synthetic.superpoint_state_dict(0),
                                               "lightglue": synthetic.lightglue_state_dict(0, "passthrough")}})
kw = dict(quality=matching.Quality.HIGH, tile_selection=matching.TileSelection.NONE, max_keypoints=K,
          geometric_verification=matching.GeometricVerification.NONE)
for _ in range(4):
    m.match(a, b, **kw)
eng = m.engine
sm = next(iter(eng.graphs.values()))
T = {k: [] for k in ("stack", "upload_enqueue", "replay_enqueue", "synchronize", "features_to_host x2", "matches_to_host", "whole match()")}
for r in range(12):
    t0 = time.perf_counter()
    st = np.stack([a, b]); t1 = time.perf_counter()
    sm._inp.copy_(torch.from_numpy(st), non_blocking=True); t2 = time.perf_counter()
    sm._graph.replay(); t3 = time.perf_counter()
    eng.synchronize(); t4 = time.perf_counter()
    k0, d0, s0 = eng.features_to_host(0, channels_first=True)
    k1, d1, s1 = eng.features_to_host(1, channels_first=True); t5 = time.perf_counter()
    out = eng.matches_to_host(len(k0), len(k1)); t6 = time.perf_counter()
    for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
        T[k].append(v * 1e3)
    t = time.perf_counter()
    m.match(a, b, **kw)
    T["whole match()"].append((time.perf_counter() - t) * 1e3)
for k, v in T.items():
    v = sorted(v[2:])
    print(f"{k:24s} median {v[len(v) // 2]:7.3f} ms   min {v[0]:7.3f}")
