#!/usr/bin/env python3
"""Experiment: S independent engines on S HIP streams, pairs dealt round-robin (concurrent kernels fill idle CUs)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icepy4d_amd import synthetic
from icepy4d_amd.engine import Engine
from icepy4d_amd.sequence import SequenceMatcher, new_table
H, W, K = 1080, 1920, 4096
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
sp, lg = synthetic.superpoint_state_dict(0), synthetic.lightglue_state_dict(0, "passthrough")
pool = []
for i in range(4):
    a, b = synthetic.stereo_pair(i, H, W)
    pool.append(torch.from_numpy(np.stack([a, b])).cuda())
streams = [torch.cuda.Stream() for _ in range(S)]
sms, tabs = [], []
for s in streams:
    with torch.cuda.stream(s):
        e = Engine(0); e.load_state_dict("superpoint", sp); e.load_state_dict("lightglue", lg)
        sms.append(SequenceMatcher(e, H, W, K)); tabs.append(new_table(steps, K, e.device))
        for i in range(2):
            sms[-1].match_pair(pool[i], i, tabs[-1], i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    j = i % S
    with torch.cuda.stream(streams[j]):
        sms[j].match_pair(pool[i % 4], i, tabs[j], i // S)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"streams={S}: {steps / dt:.2f} pairs/s  ({1e3 * dt / steps:.2f} ms/pair)")
