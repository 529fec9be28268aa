#!/usr/bin/env python3
"""Per-launch-class HIP-event times (im_profile_begin / end) of the 1080p / 4096-keypoint pair with P pairs per launch on ONE stream:
    python tools/profile_pairs_per_launch.py [P ...]
Shows what the batch dimension over pairs does to each kernel class (per PAIR: launch time / P)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from icepy4d_amd import synthetic
from icepy4d_amd.engine import Engine
from icepy4d_amd.sequence import SequenceMatcher, new_table

H, W, K = 1080, 1920, 4096
pairs = [torch.from_numpy(np.stack(synthetic.stereo_pair(e, H, W))).cuda() for e in range(4)]
for P in [int(x) for x in (sys.argv[1:] or ["1", "2"])]:
    e = Engine(0)
    e.load_state_dict("superpoint", synthetic.superpoint_state_dict(0))
    e.load_state_dict("lightglue", synthetic.lightglue_state_dict(0, "passthrough"))
    sm = SequenceMatcher(e, H, W, K, use_graph=False, pairs_per_launch=P)
    tab = new_table(8, K, e.device)
    for i in range(4 * P):
        sm.match_pair(pairs[i % 4], i, tab, i % 8)
    sm.flush(); torch.cuda.synchronize()
    e.ctx.call("im_profile_begin")
    n = 4 * P
    for i in range(n):
        sm.match_pair(pairs[i % 4], i, tab, i % 8)
    sm.flush(); torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    e.ctx.call("im_profile_end", buf, len(buf))
    prof = json.loads(buf.value.decode())
    cal = prof.pop("_empty_event_pair")
    ov = cal["total_ms"] / cal["count"]
    per_pair = {k: round((v["total_ms"] - v["count"] * ov) / n, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}
    print(f"P={P} ms per pair (sum {sum(per_pair.values()):.3f}):", json.dumps(per_pair), flush=True)
    e.close()
