"""One pair through the forward passes with direct launches (no graph), for rocprofv3 --pmc passes: every kernel of the path
appears under its own name in the counter CSV.   python3 tools/run_pair_once.py lightglue|superglue [reps]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icepy4d_amd import synthetic
from icepy4d_amd.engine import Engine
mode = sys.argv[1] if len(sys.argv) > 1 else "lightglue"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
e = Engine(0)
e.load_state_dict("superpoint", synthetic.superpoint_state_dict(0))
if mode == "superglue":
    H, W, K = 3000, 4000, 16384
    e.load_state_dict("superglue", synthetic.superglue_state_dict(0, "passthrough"))
    a, b = synthetic.translated_pair(5, H, W, 48, 16)
else:
    H, W, K = 1080, 1920, 4096
    e.load_state_dict("lightglue", synthetic.lightglue_state_dict(0, "passthrough"))
    a, b = synthetic.stereo_pair(0, H, W)
e.reserve(H, W, 2, K)
pair = torch.from_numpy(np.stack([a, b])).cuda()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(reps):
        if mode == "superglue":
            e.superpoint(pair, 3, 0.001, 4, K, flavour=1)
            e.superglue((H, W), (H, W), sinkhorn_iterations=20, match_threshold=0.3)
        else:
            e.superpoint(pair, 4, 0.0005, 4, K)
            e.lightglue((W, H), (W, H))
    s.synchronize()
print(mode, "n =", e.n.tolist(), "matches =", int((e.matches[0] > -1).sum()))
