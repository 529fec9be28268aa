#!/usr/bin/env python3
"""Experiment: can the extraction of one launch group (Winograd convolutions: matrix pipe 0.6 busy) share the CUs with the matching of
another (attention: 0.86 busy)? Two launch groups on two streams, optionally started HALF A GROUP APART so that one is in SuperPoint
while the other is in LightGlue; IM_ATTN_GROUPS=1 makes an attention block small enough (68 KB LDS, 4 waves) to sit next to a
convolution block (77 KB) on a CU.
    IM_ATTN_GROUPS=1 python tools/bench_costream.py <pairs_per_launch> <stagger 0|1> [steps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icepy4d_amd import synthetic
from icepy4d_amd.engine import Engine
from icepy4d_amd.sequence import PairPipeline, new_table
H, W, K = 1080, 1920, 4096
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
stagger = int(sys.argv[2]) if len(sys.argv) > 2 else 0
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 96
sp, lg = synthetic.superpoint_state_dict(0), synthetic.lightglue_state_dict(0, "passthrough")
pool = [torch.from_numpy(np.stack(synthetic.stereo_pair(i, H, W))).cuda() for i in range(4)]


def make_engine():
    e = Engine(0); e.load_state_dict("superpoint", sp); e.load_state_dict("lightglue", lg)
    return e


pipe = PairPipeline(make_engine, H, W, K, n_streams=2, use_graph=True, pairs_per_launch=P)
tab = new_table(steps + 64, K, pipe.device)


def run(n, first_row):
    for i in range(n):
        pipe.match_pair(pool[i % 4], i, tab, first_row + i)
    pipe.flush()


run(8 * P, 0)
pipe.synchronize()
if stagger:
    # one launch group on slot 0 alone, and slot 1 is only fed once slot 0 is about a third into it (its SuperPoint part)
    for i in range(P):
        pipe.match_pair(pool[i % 4], i, tab, steps + i)
    time.sleep(0.0035 * P)
t0 = time.perf_counter()
run(steps, 0)
pipe.synchronize()
dt = time.perf_counter() - t0
print(f"groups={os.environ.get('IM_ATTN_GROUPS', '2')} pairs_per_launch={P} stagger={stagger}: {steps / dt:.2f} pairs/s ({1e3 * dt / steps:.3f} ms/pair)", flush=True)
pipe.close()
