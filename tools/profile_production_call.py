#!/usr/bin/env python3
"""cProfile of the production-like tile call (`main_dev.py:115-132` parameters) to see the host-side share. GPU box only."""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icepy4d_amd import matching, synthetic

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
pr = cProfile.Profile()
pr.enable(); call(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
