#!/usr/bin/env python3
"""Config 5 timing: 12 MP pair, 16384 keypoints, SuperPoint + SuperGlue (20 Sinkhorn iterations), per-kernel breakdown."""
import ctypes, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icepy4d_amd import synthetic
from icepy4d_amd.engine import Engine
H, W, K = 3000, 4000, 16384
img0, img1 = synthetic.translated_pair(5, H, W, 48, 16)
e = Engine(0)
e.load_state_dict("superpoint", synthetic.superpoint_state_dict(0))
e.load_state_dict("superglue", synthetic.superglue_state_dict(0, "passthrough"))
e.reserve(H, W, 2, K)
pair = torch.from_numpy(np.stack([img0, img1])).cuda()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    def step():
        e.superpoint(pair, 3, 0.001, 4, K, flavour=1)
        e.superglue((H, W), (H, W), sinkhorn_iterations=20, match_threshold=0.3)
    for _ in range(2):
        step()
    s.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        step()
    s.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"config 5: {dt * 1e3:.1f} ms per pair ({1 / dt:.2f} pairs/s), n = {e.n.tolist()}, matches = {int((e.matches[0] > -1).sum())}")
    e.ctx.call("im_profile_begin")
    step()
    buf = ctypes.create_string_buffer(1 << 16)
    e.ctx.call("im_profile_end", buf, len(buf))
    prof = json.loads(buf.value.decode())
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
        print(f"  {k:22s} {v['count']:4d} launches  {v['total_ms']:9.3f} ms")
    sk = prof.get("sinkhorn", {}).get("total_ms")
    if sk:
        traffic = (2 * 20) * (K + 1) * (K + 1) * 4
        print(f"  sinkhorn: {traffic / 1e9:.1f} GB algorithmic / {sk:.2f} ms = {traffic / sk / 1e9:.2f} TB/s")
