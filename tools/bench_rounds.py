#!/usr/bin/env python3
"""Experiment: several timed rounds of the pipelined pair loop inside ONE process (is a slow first measurement a
property of the process, or of the first seconds?). Usage: python tools/bench_rounds.py [streams] [rounds] [steps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icepy4d_amd import synthetic
from icepy4d_amd.engine import Engine
from icepy4d_amd.sequence import PairPipeline, new_table
H, W, K = 1080, 1920, 4096
S = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
sp, lg = synthetic.superpoint_state_dict(0), synthetic.lightglue_state_dict(0, os.environ.get("IM_LG_VARIANT", "passthrough"))

def make_engine():
    e = Engine(0); e.load_state_dict("superpoint", sp); e.load_state_dict("lightglue", lg)
    return e

pipe = PairPipeline(make_engine, H, W, K, n_streams=S)
pool = []
for i in range(4):
    a, b = synthetic.stereo_pair(i, H, W)
    pool.append(torch.from_numpy(np.stack([a, b])).cuda())
table = new_table(steps, K, pipe.device)
for r in range(rounds):
    pipe.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        pipe.match_pair(pool[i % 4], i, table, i)
    pipe.synchronize(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"round {r}: {steps / dt:.2f} pairs/s  ({1e3 * dt / steps:.2f} ms/pair)", flush=True)
