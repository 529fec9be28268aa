#!/usr/bin/env python3
"""The `assign` launch class (lse_stats, col_lse_combine, best_sweep, col_best_combine, filter_scatter) inside a LightGlue forward of the
benchmark pair at ONE and at TEN pairs per launch, per pair, from the library's own HIP events, with a SHA-1 of the matches and matching
scores (to compare builds bit for bit: ICEMATCH_LIB selects the library). Round 5's A/B tool for the assignment sweeps.

    python tools/bench_assign_pairs.py
"""
import ctypes, json, os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from icepy4d_amd import synthetic
from icepy4d_amd.engine import Engine
eng = Engine(0)
eng.load_state_dict("superpoint", synthetic.superpoint_state_dict(0))
eng.load_state_dict("lightglue", synthetic.lightglue_state_dict(0, "passthrough"))
a, b = synthetic.stereo_pair(0, 1080, 1920)
img = torch.from_numpy(np.stack([a, b])).cuda()
for P in (1, 10):
    pairs = img.repeat(P, 1, 1).contiguous()
    eng.reserve(1080, 1920, 2 * P, 4096)
    eng.superpoint(pairs, max_kpts=4096)
    for _ in range(2):
        eng.lightglue((1920.0, 1080.0), (1920.0, 1080.0), n_pairs=P)
    torch.cuda.synchronize()
    eng.ctx.call("im_profile_begin")
    for _ in range(4):
        eng.lightglue((1920.0, 1080.0), (1920.0, 1080.0), n_pairs=P)
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    eng.ctx.call("im_profile_end", buf, len(buf))
    prof = json.loads(buf.value.decode())
    cal = prof.pop("_empty_event_pair", None)
    ov = cal["total_ms"] / cal["count"] if cal and cal["count"] else 0.0
    print(P, "pairs per launch: assign", round(1e3 * (prof["assign"]["total_ms"] - prof["assign"]["count"] * ov) / 4 / P, 1), "us per pair;",
          "matches sha", __import__("hashlib").sha1(eng.matches[:2].cpu().numpy().tobytes() + eng.mscores[:2].cpu().numpy().tobytes()).hexdigest()[:12])
