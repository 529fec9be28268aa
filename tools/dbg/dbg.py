import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np, torch
from icepy4d_amd import synthetic
from conftest import load_golden
from icepy4d_amd.matching import GeometricVerification, LightGlueMatcher, Quality, TileSelection
g = load_golden("g4_wrappers")
sds = {"superpoint": synthetic.superpoint_state_dict(0), "lightglue": synthetic.lightglue_state_dict(0, "passthrough")}
m = LightGlueMatcher({"state_dicts": sds})
for use_graph in (False, True):
    m._opt["use_graph"] = use_graph
    for (h, w) in [(119, 171), (119, 169), (100, 152), (119, 171), (200, 304)]:
        a = np.ascontiguousarray(g["image0"][:h, :w]); b = np.ascontiguousarray(g["image1"][:h, :w])
        f0, f1, m0, conf = m._match_images(a, b, max_keypoints=256)
        print(use_graph, h, w, len(f0.keypoints), int((m0 > -1).sum()), flush=True)
a = np.ascontiguousarray(g["image0"][:119, :171]); b = np.ascontiguousarray(g["image1"][:119, :169])
print(m._match_images(a, b, max_keypoints=256)[2].shape, flush=True)
cfg = dict(geometric_verification=GeometricVerification.NONE, max_keypoints=256, grid=[2, 2], overlap=20)
m2 = LightGlueMatcher({"state_dicts": sds})
m2._sp_params = lambda **config: None
m2.match(g["image0"], g["image1"], quality=Quality.HIGH, tile_selection=TileSelection.EXHAUSTIVE, **cfg)
print("m2 ok", len(m2.mkpts0))
