#!/usr/bin/env python3
"""Completion timeline of the launch groups of a short timed run (why K = 20 steps measure lower than K = 256)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icepy4d_amd import synthetic
from icepy4d_amd.engine import Engine
from icepy4d_amd.sequence import PairPipeline, new_table

sp_sd = synthetic.superpoint_state_dict(0); lg_sd = synthetic.lightglue_state_dict(0, "passthrough")
def make_engine():
    e = Engine(0); e.load_state_dict("superpoint", sp_sd); e.load_state_dict("lightglue", lg_sd); return e
sm = PairPipeline(make_engine, 1080, 1920, 4096, n_streams=2, use_graph=True, matcher="lightglue", pairs_per_launch=2)
pool = [torch.from_numpy(np.stack(synthetic.stereo_pair(e, 1080, 1920))).cuda() for e in range(4)]
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
gap_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
table = new_table(K, 4096, sm.device); scratch = new_table(16, 4096, sm.device)
for rep in range(3):
    for i in range(64):
        sm.match_pair(pool[i % 4], i, scratch, i % 16)
    sm.synchronize(); torch.cuda.synchronize()
    if gap_ms: time.sleep(gap_ms / 1e3)
    evs = []
    t0 = time.perf_counter()
    for i in range(K):
        sm.match_pair(pool[i % 4], i, table, i)
        if i % 2 == 1:
            eng, stream, s = sm.slots[(i // 2) % 2]
            ev = torch.cuda.Event(enable_timing=False); ev.record(stream); evs.append(ev)
    sm.flush()
    done = []
    for ev in evs:
        ev.synchronize(); done.append((time.perf_counter() - t0) * 1e3)
    sm.synchronize(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    print(f"K={K} gap={gap_ms} ms: total {dt:.1f} ms = {dt / K:.2f} ms/step; group completions (ms): " + " ".join(f"{d:.1f}" for d in done), flush=True)
