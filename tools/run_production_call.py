#!/usr/bin/env python3
"""The reference driver's call (`main_dev.py:115-132` parameters, as bench.py's `production_call_ms`) for a kernel trace: two warm-up calls, then
ONE call between two markers printed with its wall time, the matcher's own timer split and the tile pairs the preselection chose.
    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/run_production_call.py
`tools/summarize_production_call.py <dir>` then sums the kernels that started inside the measured call."""
import json, logging, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from icepy4d_amd import matching, synthetic

chosen = []


class _Grab(logging.Handler):
    def emit(self, rec):
        msg = rec.getMessage()
        if "tile pairs" in msg.lower() and msg.strip() not in chosen:
            chosen.append(msg.strip())


logging.getLogger().addHandler(_Grab())
for name in list(logging.root.manager.loggerDict):
    if name.startswith("icepy4d_amd"):
        logging.getLogger(name).setLevel(logging.INFO)

m = matching.LightGlueMatcher({"state_dicts": {"superpoint": synthetic.superpoint_state_dict(0),
                                               "lightglue": synthetic.lightglue_state_dict(0, "passthrough")}})
ha, hb = synthetic.translated_pair(3, 1000, 1500, 24, 8, noise=0.0)
a3 = np.repeat(np.kron(ha, np.ones((4, 4), np.uint8))[:, :, None], 3, 2)
b3 = np.repeat(np.kron(hb, np.ones((4, 4), np.uint8))[:, :, None], 3, 2)


def call():
    m.match(a3, b3, quality=matching.Quality.HIGH, tile_selection=matching.TileSelection.PRESELECTION, grid=[2, 2], overlap=200,
            origin=[0, 0], min_matches_per_tile=3, max_keypoints=8196,
            geometric_verification=matching.GeometricVerification.PYDEGENSAC, threshold=2, confidence=0.9999)


call(); call()
torch.cuda.synchronize()
chosen.clear()
# a marker kernel on each side of the measured call (erfinv: nothing in the call uses it): the summary cuts the trace at its last two launches
mark = torch.zeros(1 << 10, device="cuda")
mark.erfinv_(); torch.cuda.synchronize()
t0 = time.perf_counter()
call()
torch.cuda.synchronize()
wall = 1e3 * (time.perf_counter() - t0)
mark.erfinv_(); torch.cuda.synchronize()
print(json.dumps({"wall_ms": round(wall, 2), "timer_split_ms": {k: round(1e3 * v, 2) for k, v in m.timer.times.items()}, "matched_points": int(len(m.mkpts0)),
                  "tile_pairs_log": chosen}))
