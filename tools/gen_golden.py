#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE modules (imported from /root/reference,
build container only) on seeded weights and inputs.  Nothing of the reference travels: only the
input/output vectors written here are committed.  Re-run: `python tools/gen_golden.py` (every fixture; `--out DIR` writes
elsewhere, e.g. to compare with the committed files).

Un-vendored dependencies of the reference that are absent here (cv2, kornia, easydict, tkinter) are
replaced by empty stub modules; no code path that would *call* them is exercised (gray uint8 inputs,
Quality.HIGH, GeometricVerification.NONE).  Weight loaders (`torch.hub.load_state_dict_from_url`,
`torch.load` of the stripped .pth files) are patched to return nothing, then the seeded state dicts of
`icepy4d_amd.synthetic` are installed with `load_state_dict`.
"""
import hashlib
import logging
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF_SRC = "/root/reference/src"
OUT = os.path.join(ROOT, "tests", "golden")

from icepy4d_amd import synthetic  # noqa: E402


def install_stubs():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    stub("cv2")
    k = stub("kornia")
    k.feature = stub("kornia.feature", DISK=object)
    k.color = stub("kornia.color")
    k.geometry = stub("kornia.geometry")

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for key, v in {**(d or {}), **kw}.items():
                self[key] = v

        def __setitem__(self, key, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            super().__setitem__(key, v)

        def __getattr__(self, key):
            try:
                return self[key]
            except KeyError:
                raise AttributeError(key)

        __setattr__ = __setitem__

    stub("easydict", EasyDict=EasyDict)
    import matplotlib
    matplotlib.use = lambda *a, **k: None
    torch.hub.load_state_dict_from_url = lambda *a, **k: None
    _load = torch.load
    torch.load = lambda f, *a, **k: None if str(f).endswith(".pth") else _load(f, *a, **k)
    _lsd = torch.nn.Module.load_state_dict

    def load_state_dict(self, sd, *a, **k):
        return None if sd is None else _lsd(self, sd, *a, **k)

    torch.nn.Module.load_state_dict = load_state_dict


def sha(t) -> str:
    a = t.detach().cpu().contiguous().numpy() if isinstance(t, torch.Tensor) else np.ascontiguousarray(t)
    return hashlib.sha256(a.tobytes()).hexdigest()


def npy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in arrays.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def noise_image(seed, h, w):
    return synthetic.band_limited_noise(np.random.default_rng(seed), h, w)


def kornia_rgb_to_grayscale(image, rgb_weights=None):
    """RESTATEMENT of kornia.color.rgb_to_grayscale for floating-point images (kornia is un-vendored and absent here):
    weights (0.299, 0.587, 0.114) in the image dtype, `w_r * r + w_g * g + w_b * b`. Parity unpinned at this call site."""
    r, g, b = image[..., 0:1, :, :], image[..., 1:2, :, :], image[..., 2:3, :, :]
    w = torch.tensor([0.299, 0.587, 0.114], device=image.device, dtype=image.dtype)
    return w[0] * r + w[1] * g + w[2] * b


def cv2_cvtcolor_rgb2gray(image, code=None):
    """RESTATEMENT of cv2.cvtColor(img, cv2.COLOR_RGB2GRAY) for uint8 (OpenCV is un-vendored and absent here): fixed point
    with 14 fractional bits, (4899 R + 9617 G + 1868 B + 8192) >> 14. Parity unpinned at this call site."""
    a = image.astype(np.uint32)
    return ((a[..., 0] * 4899 + a[..., 1] * 9617 + a[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)


def gen_colour(sp_sd):
    """G6: 3-channel uint8 input (what `core/images.py:75` hands to `match()`) through the reference's own code paths:
    LightGlue flavour = `_frame2tensor` + `SuperPoint.extract` (`matchers.py:1212-1220`, `lightglue/superpoint.py:217-231`,
    `lightglue/utils.py:35-36`), SuperGlue flavour = `SuperGlueMatcher._match_images` (`matchers.py:911-917`). The two
    un-vendored colour conversions are the restatements above, installed into the stub modules."""
    from PIL import Image
    from icepy4d.thirdparty.LightGlue.lightglue import superpoint as r_sp
    from icepy4d.thirdparty.SuperGlue.models import superpoint as r_sgsp
    from icepy4d.matching import matchers as r_m
    sys.modules["kornia"].color.rgb_to_grayscale = kornia_rgb_to_grayscale
    sys.modules["cv2"].cvtColor = cv2_cvtcolor_rgb2gray
    sys.modules["cv2"].COLOR_RGB2GRAY = 7
    rgb = np.asarray(Image.open("/root/reference/assets/img/cam1/IMG_2637.jpg").convert("RGB"))
    rgb = np.ascontiguousarray(rgb[300:500, 400:704])                       # 200 x 304 x 3 crop of a real colour image
    lgm = r_m.LightGlueMatcher.__new__(r_m.LightGlueMatcher)               # only its _frame2tensor is needed
    x = r_m.LightGlueMatcher._frame2tensor(lgm, rgb, "cpu")
    ext = r_sp.SuperPoint(max_num_keypoints=300).eval()
    ext.load_state_dict(sp_sd)
    sg_net = r_sgsp.SuperPoint({"nms_radius": 3, "keypoint_threshold": 0.001, "max_keypoints": 300}).eval()
    sg_net.load_state_dict(sp_sd)
    with torch.inference_mode():
        out = ext.extract(x, resize=None)
        gray_u8 = sys.modules["cv2"].cvtColor(rgb, 7)                       # `matchers.py:911-912`
        t = torch.from_numpy(gray_u8 / 255.0).float()[None, None]          # base `_frame2tensor` (`matchers.py:263-274`)
        sg_out = sg_net({"image": t})
    print(f"  colour: LG kpts {out['keypoints'].shape[1]}, SG kpts {sg_out['keypoints'][0].shape[0]}")
    save("g6_colour", rgb=rgb, lg_keypoints=out["keypoints"][0], lg_scores=out["keypoint_scores"][0],
         lg_descriptors=out["descriptors"][0], lg_gray=kornia_rgb_to_grayscale(x[None])[0, 0],
         sg_gray_u8=gray_u8, sg_keypoints=sg_out["keypoints"][0], sg_scores=sg_out["scores"][0],
         sg_descriptors=sg_out["descriptors"][0])


def gen_features_pickle():
    """g7: a `Features` pickle written by the reference's own class (`core/features.py:362-453, 596-600`) from seeded arrays,
    the file format the epoch loop persists (`main_dev.py:160-173`)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("icepy4d.core.features", REF_SRC + "/icepy4d/core/features.py")
    mod = importlib.util.module_from_spec(spec)
    for name in ("icepy4d", "icepy4d.core"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["icepy4d.core.features"] = mod
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(77)
    n = 12
    kpts = rng.integers(4, 600, size=(n, 2)).astype(np.float32)
    descr = rng.normal(size=(256, n)).astype(np.float32)
    scores = rng.uniform(0.01, 0.5, size=n).astype(np.float32)
    fs = mod.Features()
    fs.append_features_from_numpy(x=kpts[:, 0], y=kpts[:, 1], descr=descr, scores=scores)
    path = os.path.join(OUT, "g7_features_ref.pkl")
    fs.save_as_pickle(path)
    fs2 = mod.Features()
    fs2.append_features_from_numpy(x=kpts[:, 0], y=kpts[:, 1], descr=descr, scores=scores.reshape(-1, 1), epoch=3)
    fs2.save_as_pickle(os.path.join(OUT, "g7_features_ref_epoch.pkl"))
    save("g7_features_in", kpts=kpts, descr=descr, scores=scores)
    print(f"wrote {path} ({os.path.getsize(path)} bytes)")
    for name in ("icepy4d.core.features", "icepy4d.core", "icepy4d"):
        if isinstance(sys.modules.get(name), types.ModuleType) and not getattr(sys.modules[name], "__file__", None):
            sys.modules.pop(name, None)
    sys.modules.pop("icepy4d.core.features", None)


def gen_triangulation():
    """G10 (row f-4): the reference's `triangulate_points_linear` / `triangulate_nviews` (`sfm/triangulation.py:153-186`, pure numpy) on seeded
    cameras and 500 noisy correspondences, plus a three-view point. The module is loaded from the reference tree with its sibling imports
    (cv2-based geometry, colour interpolation, Camera, the thirdparty LS triangulation: none of them used by the two functions) stubbed."""
    import importlib.util
    saved = {k: sys.modules.get(k) for k in ("icepy4d", "icepy4d.sfm", "icepy4d.sfm.geometry", "icepy4d.sfm.interpolate_colors", "icepy4d.core",
                                             "icepy4d.core.camera", "icepy4d.utils", "icepy4d.utils.math", "icepy4d.thirdparty",
                                             "icepy4d.thirdparty.triangulation", "icepy4d.sfm.triangulation")}

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__path__ = []
        sys.modules[name] = m
        return m

    stub("icepy4d"); stub("icepy4d.sfm"); stub("icepy4d.core"); stub("icepy4d.utils"); stub("icepy4d.thirdparty")
    stub("icepy4d.sfm.geometry", undistort_points=None)
    stub("icepy4d.sfm.interpolate_colors", interpolate_point_colors=None)
    stub("icepy4d.core.camera", Camera=object)
    stub("icepy4d.utils.math", convert_from_homogeneous=None, convert_to_homogeneous=None)
    stub("icepy4d.thirdparty.triangulation", iterative_LS_triangulation=None)
    spec = importlib.util.spec_from_file_location("icepy4d.sfm.triangulation", REF_SRC + "/icepy4d/sfm/triangulation.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["icepy4d.sfm.triangulation"] = mod
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(1010)
    n = 500
    X = np.c_[rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(6, 14, n)]
    K0 = np.array([[6621.7, 0, 3000.4], [0, 6620.1, 1998.7], [0, 0, 1]])
    K1 = np.array([[6588.3, 0, 3012.9], [0, 6590.6, 2005.2], [0, 0, 1]])

    def rot(ax, ay, az):
        cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
        return (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
                @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))

    R1, t1 = rot(0.02, 0.21, -0.015), np.array([-2.1, 0.07, 0.3])
    R2, t2 = rot(-0.03, -0.17, 0.01), np.array([1.8, -0.1, 0.25])
    P0, P1, P2 = K0 @ np.eye(3, 4), K1 @ np.c_[R1, t1], K0 @ np.c_[R2, t2]
    h = np.c_[X, np.ones(n)]

    def project(P):
        x = (P @ h.T).T
        return x[:, :2] / x[:, 2:]

    x0 = project(P0) + rng.normal(0, 0.4, (n, 2))
    x1 = project(P1) + rng.normal(0, 0.4, (n, 2))
    x2 = project(P2) + rng.normal(0, 0.4, (n, 2))
    h0, h1, h2 = np.c_[x0, np.ones(n)], np.c_[x1, np.ones(n)], np.c_[x2, np.ones(n)]
    X01 = mod.triangulate_points_linear(P0, P1, h0, h1)
    X3 = np.array([mod.triangulate_nviews([P0, P1, P2], [a, b, c]) for a, b, c in zip(h0[:50], h1[:50], h2[:50])])
    print(f"  triangulation: two views {X01.shape}, median error {np.median(np.linalg.norm(X01[:, :3] - X, axis=1)):.4f}; three views {X3.shape}")
    save("g10_triangulation", P0=P0, P1=P1, P2=P2, x0=h0, x1=h1, x2=h2, points_true=X, X_two_views=X01, X_three_views=X3)
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def gen_prune_threshold():
    """g2_lightglue_6: the reference's `desc.shape[-2] > pruning_th` gate (`lightglue.py:495, 503`). On a CPU tensor the
    reference looks up `pruning_keypoint_thresholds['cpu']` = -1; its CUDA values are 1024 / 1536. The table entry is set to
    280 here so that the gate is exercised at fixture size (image 0: 300 points, pruned until <= 280 are left; image 1: 257
    points, never pruned) by the reference's own control flow."""
    from icepy4d.thirdparty.LightGlue.lightglue import lightglue as r_lg
    variant, m, n = "prune", 300, 257
    sd = synthetic.lightglue_state_dict(0, variant)
    net = r_lg.LightGlue(features="superpoint", depth_confidence=-1).eval()
    net.load_state_dict(sd)
    net.pruning_keypoint_thresholds = dict(net.pruning_keypoint_thresholds, cpu=280)
    f = synthetic.synthetic_features(4, m, n)
    data = {"image0": {"keypoints": torch.from_numpy(f["kpts0"])[None], "descriptors": torch.from_numpy(f["desc0"])[None],
                       "image_size": torch.from_numpy(f["size0"])[None]},
            "image1": {"keypoints": torch.from_numpy(f["kpts1"])[None], "descriptors": torch.from_numpy(f["desc1"])[None],
                       "image_size": torch.from_numpy(f["size1"])[None]}}
    with torch.inference_mode():
        out = net(data)
    print(f"  LG prune-threshold case: matches={int((out['matches0'] > -1).sum())} prune0 max {int(out['prune0'].max())} "
          f"prune1 max {int(out['prune1'].max())}")
    save("g2_lightglue_6", variant=variant, m=m, n=n, depth_confidence=-1, width_confidence=0.99, seed=4, pruning_min_kpts=280,
         matches0=out["matches0"][0], matches1=out["matches1"][0], matching_scores0=out["matching_scores0"][0],
         matching_scores1=out["matching_scores1"][0], stop=out["stop"], prune0=out["prune0"][0], prune1=out["prune1"][0])


G9_CASES = (("prune_gradual", 2048, 1536, {}), ("prune_gradual", 2048, 1536, {"depth_confidence": -1}),
            ("earlystop_late", 2048, 1536, {}), ("prune_gradual", 1536, 2048, {"width_confidence": 0.95}),
            ("prune_gradual", 4096, 3000, {}), ("prune_gradual", 2500, 4096, {"depth_confidence": -1}))


def gen_adaptive_large():
    """g9_lightglue_adaptive_{0..5}: the reference's `LightGlue` (`lightglue/lightglue.py:436-556`) on synthetic features of 2048 / 1536
    (and 4096 / 3000, 2500 / 4096) points with weights under which the adaptive machinery is at work in EVERY layer, the way it is with trained weights on the CPU
    path (pruning evaluated after every layer, `:326-331, 495-510`): live widths walk through (1024, 2048] and far below on both
    images, the pair stops on its own once the pruned points lift the confident ratio over 0.95 (`:571-579`) - or runs all nine
    layers with depth_confidence=-1, where `get_pruning_mask` has no token confidences (`:563-569`) - and `earlystop_late` stops at
    layer 7 at full width. Outputs only (matches, scores, stop, prune counters = the layer each point was dropped at)."""
    from icepy4d.thirdparty.LightGlue.lightglue import lightglue as r_lg
    for ci, (variant, m, n, conf) in enumerate(G9_CASES):
        sd = synthetic.lightglue_state_dict(0, variant)
        net = r_lg.LightGlue(features="superpoint", **conf).eval()
        net.load_state_dict(sd)
        f = synthetic.synthetic_features(90 + ci, m, n)
        data = {"image0": {"keypoints": torch.from_numpy(f["kpts0"])[None], "descriptors": torch.from_numpy(f["desc0"])[None],
                           "image_size": torch.from_numpy(f["size0"])[None]},
                "image1": {"keypoints": torch.from_numpy(f["kpts1"])[None], "descriptors": torch.from_numpy(f["desc1"])[None],
                           "image_size": torch.from_numpy(f["size1"])[None]}}
        with torch.inference_mode():
            out = net(data)
        p0, p1 = out["prune0"][0], out["prune1"][0]
        live = [[int((p0 > l).sum()), int((p1 > l).sum())] for l in range(int(out["stop"]))]
        print(f"  LG adaptive case {ci} {variant} {m}x{n} {conf}: stop={out['stop']} matches={int((out['matches0'] > -1).sum())} live per layer {live}")
        save(f"g9_lightglue_adaptive_{ci}", variant=variant, m=m, n=n, seed=90 + ci,
             depth_confidence=conf.get("depth_confidence", 0.95), width_confidence=conf.get("width_confidence", 0.99),
             matches0=out["matches0"][0], matches1=out["matches1"][0],
             matching_scores0=out["matching_scores0"][0], matching_scores1=out["matching_scores1"][0],
             stop=out["stop"], matches=out["matches"][0], scores=out["scores"][0], prune0=p0, prune1=p1, live=np.array(live))


# G8's input: a pair whose HALF-resolution pyramid level is textured and related by a translation (each pixel of a half-size
# translated pair blown up to a 2 x 2 block: with seeded weights the doubly smoothed full-size noise would leave only a few dozen
# low-resolution matches). Parameters found by `python tools/gen_golden.py preselection_search`: with them some tile pair holds 4 or
# 5 preselection matches, so the reference's selection (quirk q2: threshold always 5) differs from what min_matches_per_tile=3 asks.
G8_PARAMS = dict(seed=21, half_h=200, half_w=304, dx=8, dy=8, grid=[2, 2], overlap=11, min_matches_per_tile=3, max_keypoints=512)


def g8_pair(seed, half_h, half_w, dx, dy, **_):
    ha, hb = synthetic.translated_pair(seed, half_h, half_w, dx, dy, noise=0.0)
    return np.kron(ha, np.ones((2, 2), np.uint8)), np.kron(hb, np.ones((2, 2), np.uint8))


def run_reference_preselection(sp_sd, lg_sd, params):
    """The reference's own `LightGlueMatcher.match(..., tile_selection=PRESELECTION, grid=, overlap=, min_matches_per_tile=)`
    (`matchers.py:139-261, 304-469, 471-581`; the call of `main_dev.py:115-132`) on seeded weights. `cv2.pyrDown` - un-vendored,
    absent here - is `oracle/pyramid_cpu.pyr_down` installed into the stub module, as G6 does for `cvtColor`. Returns the arrays of
    the fixture: what `_tile_selection` returned, the preselection matches per tile pair, and the final result properties."""
    from itertools import product
    from icepy4d.thirdparty.LightGlue.lightglue import superpoint as r_sp, lightglue as r_lg
    from icepy4d.matching import matchers as r_m
    from icepy4d.matching.enums import GeometricVerification, Quality, TileSelection
    from oracle.pyramid_cpu import pyr_down
    sys.modules["cv2"].pyrDown = pyr_down
    o_sp, o_lg = r_sp.SuperPoint.__init__, r_lg.LightGlue.__init__
    o_sel, o_mi = r_m.LightGlueMatcher._tile_selection, r_m.LightGlueMatcher._match_images
    seen = {"pairs": None, "calls": []}

    def sp_init(self, **conf):
        o_sp(self, **conf)
        self.load_state_dict(sp_sd)

    def lg_init(self, *a, **k):
        o_lg(self, *a, **k)
        self.load_state_dict(lg_sd)

    def sel(self, *a, **k):
        seen["pairs"] = o_sel(self, *a, **k)
        return seen["pairs"]

    def mi(self, i0, i1, **k):
        out = o_mi(self, i0, i1, **k)
        seen["calls"].append((i0.shape, i1.shape, out))
        return out

    r_sp.SuperPoint.__init__, r_lg.LightGlue.__init__ = sp_init, lg_init
    r_m.LightGlueMatcher._tile_selection, r_m.LightGlueMatcher._match_images = sel, mi
    import builtins
    _print = builtins.print
    builtins.print = lambda *a, **k: None
    img0, img1 = g8_pair(**params)
    try:
        m = r_m.LightGlueMatcher({"force_cpu": True})
        m.match(img0, img1, quality=Quality.HIGH, tile_selection=TileSelection.PRESELECTION, grid=params["grid"],
                overlap=params["overlap"], min_matches_per_tile=params["min_matches_per_tile"],
                geometric_verification=GeometricVerification.NONE, max_keypoints=params["max_keypoints"],
                save_dir="/tmp/gen_golden_g8")
    finally:
        builtins.print = _print
        r_sp.SuperPoint.__init__, r_lg.LightGlue.__init__ = o_sp, o_lg
        r_m.LightGlueMatcher._tile_selection, r_m.LightGlueMatcher._match_images = o_sel, o_mi
    # matches of the low-resolution call per tile pair, by the rule of `matchers.py:548-558` (diagnostic columns of the fixture:
    # they show which pairs sit between the two thresholds)
    f0, f1, mtc, _ = seen["calls"][0][2]
    n_down = 3 if img0.shape[0] > 4000 else 2 if img0.shape[0] > 2000 else 1
    v = mtc > -1
    kp0, kp1 = f0.keypoints[v] * 2 ** n_down, f1.keypoints[mtc[v]] * 2 ** n_down
    t = r_m.Tiler(grid=params["grid"], overlap=params["overlap"], origin=[0, 0])
    l0, _ = t.compute_limits_by_grid(img0)
    l1, _ = t.compute_limits_by_grid(img1)
    counts = []
    for a, b in sorted(product(l0.keys(), l1.keys())):
        r0, r1 = np.asarray(l0[a]), np.asarray(l1[b])
        ins = (np.all(kp0 > r0[:2], 1) & np.all(kp0 < r0[2:], 1)) & (np.all(kp1 > r1[:2], 1) & np.all(kp1 < r1[2:], 1))
        counts.append((a, b, int(ins.sum())))
    counts = np.array(counts, dtype=np.int64)
    assert [tuple(r[:2]) for r in counts if r[2] > 5] == [tuple(p) for p in seen["pairs"]]      # q2: the threshold in force is 5
    return dict(image0=img0, image1=img1, grid=np.array(params["grid"]), overlap=params["overlap"],
                min_matches_per_tile=params["min_matches_per_tile"], max_keypoints=params["max_keypoints"], n_down=n_down,
                tile_pairs=np.array(seen["pairs"], dtype=np.int64).reshape(-1, 2), preselection_counts=counts,
                presel_n_matches=int(v.sum()), mkpts0=m.mkpts0, mkpts1=m.mkpts1, descriptors0=m.descriptors0,
                descriptors1=m.descriptors1, scores0=m.scores0, scores1=m.scores1, mconf=m.mconf)


def gen_preselection(sp_sd):
    """G8: the production call (`main_dev.py:115-132`: PRESELECTION, a grid, an overlap, `min_matches_per_tile=3`) through the
    reference's own `match()`; pins quirk q2 (`matchers.py:353-355, 502`: the option is never found, the threshold is always 5)."""
    res = run_reference_preselection(sp_sd, synthetic.lightglue_state_dict(0, "passthrough"), G8_PARAMS)
    c = res["preselection_counts"]
    print(f"  preselection: {res['presel_n_matches']} low-resolution matches; tile pairs selected {len(res['tile_pairs'])} of {len(c)}, "
          f"with 4-5 matches (selected only if the option were honoured): {[tuple(int(x) for x in r[:2]) for r in c if 3 < r[2] <= 5]}; "
          f"final matches {len(res['mkpts0'])}")
    save("g8_preselection", **res)


def preselection_search(sp_sd):
    """Parameter search behind G8_PARAMS (not part of fixture generation): first (dx, dy, overlap) for which some tile pair holds
    4 or 5 preselection matches while the call still selects several pairs."""
    lg_sd = synthetic.lightglue_state_dict(0, "passthrough")
    for ov in (9, 10, 11, 12, 14):
        for dx, dy in ((8, 0), (8, 8), (16, 8)):
            p = dict(G8_PARAMS, dx=dx, dy=dy, overlap=ov)
            res = run_reference_preselection(sp_sd, lg_sd, p)
            c = res["preselection_counts"]
            mid = [tuple(int(x) for x in r) for r in c if 3 < r[2] <= 5]
            print(f"overlap {ov} d=({dx},{dy}): matches {res['presel_n_matches']} counts {c[:, 2].tolist()} between {mid} final {len(res['mkpts0'])}", flush=True)


def main():
    global OUT
    if "--out" in sys.argv:     # e.g. `python tools/gen_golden.py --out /tmp/golden` to compare with the committed fixtures
        i = sys.argv.index("--out")
        OUT = os.path.abspath(sys.argv[i + 1])
        del sys.argv[i:i + 2]
        os.makedirs(OUT, exist_ok=True)
    logging.disable(logging.CRITICAL)
    install_stubs()
    sys.path.insert(0, REF_SRC)
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "colour":
        gen_colour(synthetic.superpoint_state_dict(0))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "features_pickle":
        gen_features_pickle()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "triangulation":
        gen_triangulation()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "prune_threshold":
        gen_prune_threshold()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "adaptive_large":
        gen_adaptive_large()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "preselection":
        gen_preselection(synthetic.superpoint_state_dict(0))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "preselection_search":
        preselection_search(synthetic.superpoint_state_dict(0))
        return
    from icepy4d.thirdparty.LightGlue.lightglue import superpoint as r_sp, lightglue as r_lg
    from icepy4d.thirdparty.SuperGlue.models import superpoint as r_sgsp, superglue as r_sg

    sp_sd = synthetic.superpoint_state_dict(0)

    # ---------------- G1: SuperPoint, stage by stage, both flavours ----------------
    # a, b: the sizes of rounds 1-3 (kept); c: the 240 x 320 image SURVEY 8c planned (several 4-wave conv block columns per row)
    for tag, (h, w), k, seed in (("a", (96, 128), 64, 11), ("b", (136, 200), 2000, 12), ("c", (240, 320), 512, 13)):
        img = noise_image(seed, h, w)
        x = torch.tensor(img / 255.0, dtype=torch.float)[None, None]
        net = r_sp.SuperPoint(max_num_keypoints=k).eval()
        net.load_state_dict(sp_sd)
        with torch.inference_mode():
            feat = x
            for nm in ("conv1a", "conv1b"):
                feat = net.relu(getattr(net, nm)(feat))
            feat = net.pool(feat)
            for nm in ("conv2a", "conv2b"):
                feat = net.relu(getattr(net, nm)(feat))
            feat = net.pool(feat)
            for nm in ("conv3a", "conv3b"):
                feat = net.relu(getattr(net, nm)(feat))
            feat = net.pool(feat)
            for nm in ("conv4a", "conv4b"):
                feat = net.relu(getattr(net, nm)(feat))
            sc = net.convPb(net.relu(net.convPa(feat)))
            sc = torch.nn.functional.softmax(sc, 1)[:, :-1]
            b, _, hc, wc = sc.shape
            sc = sc.permute(0, 2, 3, 1).reshape(b, hc, wc, 8, 8).permute(0, 1, 3, 2, 4).reshape(b, hc * 8, wc * 8)
            nms4 = r_sp.simple_nms(sc, 4)
            nms3 = r_sgsp.simple_nms(sc, 3)
            dense = torch.nn.functional.normalize(net.convDb(net.relu(net.convDa(feat))), p=2, dim=1)
            out = net.extract(x[0], resize=None)
            sg_net = r_sgsp.SuperPoint({"nms_radius": 3, "keypoint_threshold": 0.001, "max_keypoints": {"a": 50, "b": -1, "c": 300}[tag]}).eval()
            sg_net.load_state_dict(sp_sd)
            sg_out = sg_net({"image": x})
        save(f"g1_superpoint_{tag}", image=img, max_k=k,
             feat_sha=sha(feat), feat_sample=feat[0, ::16, ::3, ::5],
             score_map=sc[0], nms4=nms4[0], nms3=nms3[0],
             dense_sha=sha(dense), dense_sample=dense[0, ::32],
             keypoints=out["keypoints"][0], keypoint_scores=out["keypoint_scores"][0],
             descriptors=out["descriptors"][0], image_size=out["image_size"][0],
             sg_keypoints=sg_out["keypoints"][0], sg_scores=sg_out["scores"][0], sg_descriptors=sg_out["descriptors"][0])

    # ---------------- G2: LightGlue on synthetic features ----------------
    cases = [("default", 128, 128, {}), ("passthrough", 128, 128, {}), ("passthrough", 300, 257, {}),
             ("earlystop", 128, 128, {}), ("prune", 300, 257, {"depth_confidence": -1}),
             ("passthrough", 96, 160, {"width_confidence": -1, "depth_confidence": -1})]
    for ci, (variant, m, n, conf) in enumerate(cases):
        sd = synthetic.lightglue_state_dict(0, variant)
        net = r_lg.LightGlue(features="superpoint", **conf).eval()
        net.load_state_dict(sd)
        f = synthetic.synthetic_features(ci, m, n)
        data = {"image0": {"keypoints": torch.from_numpy(f["kpts0"])[None], "descriptors": torch.from_numpy(f["desc0"])[None],
                           "image_size": torch.from_numpy(f["size0"])[None]},
                "image1": {"keypoints": torch.from_numpy(f["kpts1"])[None], "descriptors": torch.from_numpy(f["desc1"])[None],
                           "image_size": torch.from_numpy(f["size1"])[None]}}
        with torch.inference_mode():
            out = net(data)
            # layer-0 descriptors and the layer-0 assignment for stage tests
            e0 = net.posenc(r_lg.normalize_keypoints(data["image0"]["keypoints"], data["image0"]["image_size"]))
            e1 = net.posenc(r_lg.normalize_keypoints(data["image1"]["keypoints"], data["image1"]["image_size"]))
            s0 = net.transformers[0].self_attn(data["image0"]["descriptors"], e0)
            s1 = net.transformers[0].self_attn(data["image1"]["descriptors"], e1)
            c0, c1 = net.transformers[0].cross_attn(s0, s1)
            sc0, sim0 = net.log_assignment[0](c0, c1)
        print(f"  LG case {ci} {variant} {m}x{n}: stop={out['stop']} matches={int((out['matches0'] > -1).sum())} "
              f"pruned0={int((out['prune0'] < out['stop']).sum()) if conf.get('width_confidence', 1) > 0 else 0}")
        save(f"g2_lightglue_{ci}", variant=variant, m=m, n=n,
             depth_confidence=conf.get("depth_confidence", 0.95), width_confidence=conf.get("width_confidence", 0.99),
             seed=ci, encoding0=e0[:, 0, 0], self0=s0[0], cross0=c0[0], cross1=c1[0],
             sim_l0=sim0[0], scores_l0=sc0[0],
             matches0=out["matches0"][0], matches1=out["matches1"][0],
             matching_scores0=out["matching_scores0"][0], matching_scores1=out["matching_scores1"][0],
             stop=out["stop"], matches=out["matches"][0], scores=out["scores"][0],
             prune0=out["prune0"][0], prune1=out["prune1"][0])

    # ---------------- G3: SuperGlue on synthetic features ----------------
    for ci, (variant, m, n, iters) in enumerate((("default", 128, 128, 20), ("passthrough", 200, 150, 20),
                                                 ("passthrough", 128, 128, 100))):
        sd = synthetic.superglue_state_dict(0, variant)
        net = r_sg.SuperGlue({"sinkhorn_iterations": iters, "match_threshold": 0.3, "weights": "outdoor"}).eval()
        net.load_state_dict(sd)
        f = synthetic.synthetic_features(100 + ci, m, n)
        data = {"keypoints0": torch.from_numpy(f["kpts0"])[None], "keypoints1": torch.from_numpy(f["kpts1"])[None],
                "scores0": torch.from_numpy(f["scores0"])[None], "scores1": torch.from_numpy(f["scores1"])[None],
                "descriptors0": torch.from_numpy(f["desc0"].T.copy())[None], "descriptors1": torch.from_numpy(f["desc1"].T.copy())[None],
                "image0": torch.zeros(1, 1, 480, 640), "image1": torch.zeros(1, 1, 480, 640)}
        with torch.inference_mode():
            out = net(data)
            kn0 = r_sg.normalize_keypoints(data["keypoints0"], data["image0"].shape)
            kenc0 = data["descriptors0"] + net.kenc(kn0, data["scores0"])
            rs = np.random.default_rng(77 + ci)
            zin = torch.from_numpy(rs.normal(0, 2, size=(1, m, n)).astype(np.float32))
            ot = r_sg.log_optimal_transport(zin, net.bin_score, iters)
        print(f"  SG case {ci} {variant} {m}x{n} it={iters}: matches={int((out['matches0'] > -1).sum())}")
        save(f"g3_superglue_{ci}", variant=variant, m=m, n=n, iters=iters, seed=100 + ci,
             kenc0=kenc0[0], ot_in=zin[0], ot_out=ot[0],
             matches0=out["matches0"][0], matches1=out["matches1"][0],
             matching_scores0=out["matching_scores0"][0], matching_scores1=out["matching_scores1"][0])

    # ---------------- G4: the wrapper classes (tile modes, quirks q1/q4/q5/q6) ----------------
    from icepy4d.matching import matchers as r_m
    from icepy4d.matching.enums import GeometricVerification, Quality, TileSelection
    lg_sd = synthetic.lightglue_state_dict(0, "passthrough")
    sg_sd = synthetic.superglue_state_dict(0, "passthrough")
    _orig_sp_init = r_sp.SuperPoint.__init__
    _orig_lg_init = r_lg.LightGlue.__init__

    def sp_init(self, **conf):
        _orig_sp_init(self, **conf)
        self.load_state_dict(sp_sd)

    def lg_init(self, *a, **k):
        _orig_lg_init(self, *a, **k)
        self.load_state_dict(lg_sd)

    r_sp.SuperPoint.__init__ = sp_init
    r_lg.LightGlue.__init__ = lg_init
    import builtins
    _print = builtins.print
    img0, img1 = synthetic.translated_pair(3, 200, 304)
    cfg = dict(geometric_verification=GeometricVerification.NONE, max_keypoints=256)
    res = {}
    builtins.print = lambda *a, **k: None
    try:
        m = r_m.LightGlueMatcher({"force_cpu": True})
        m.match(img0, img1, quality=Quality.HIGH, tile_selection=TileSelection.NONE, **cfg)
        res.update(lg_none_mkpts0=m.mkpts0, lg_none_mkpts1=m.mkpts1, lg_none_desc0=m.descriptors0,
                   lg_none_scores0=m.scores0, lg_none_mconf=m.mconf)
        m = r_m.LightGlueMatcher({"force_cpu": True})
        m.match(img0, img1, quality=Quality.HIGH, tile_selection=TileSelection.GRID, grid=[2, 2], overlap=20,
                save_dir="/tmp/gen_golden_lg", **cfg)
        res.update(lg_grid_mkpts0=m.mkpts0, lg_grid_mkpts1=m.mkpts1, lg_grid_desc0=m.descriptors0, lg_grid_desc1=m.descriptors1,
                   lg_grid_scores0=m.scores0, lg_grid_scores1=m.scores1, lg_grid_mconf=m.mconf)
        sgm = r_m.SuperGlueMatcher({"weights": "outdoor", "keypoint_threshold": 0.001, "max_keypoints": 256,
                                    "match_threshold": 0.3, "force_cpu": True})
        sgm.matcher.superpoint.load_state_dict(sp_sd)
        sgm.matcher.superglue.load_state_dict(sg_sd)
        sgm.match(img0, img1, quality=Quality.HIGH, tile_selection=TileSelection.NONE,
                  geometric_verification=GeometricVerification.NONE)
        res.update(sg_none_mkpts0=sgm.mkpts0, sg_none_mkpts1=sgm.mkpts1, sg_none_desc0=sgm.descriptors0,
                   sg_none_scores0=sgm.scores0, sg_none_mconf=sgm.mconf)
        sgm.reset()
        sgm.match(img0, img1, quality=Quality.HIGH, tile_selection=TileSelection.EXHAUSTIVE, grid=[1, 2], overlap=10,
                  geometric_verification=GeometricVerification.NONE, save_dir="/tmp/gen_golden_sg")
        res.update(sg_exh_mkpts0=sgm.mkpts0, sg_exh_mkpts1=sgm.mkpts1, sg_exh_scores0=sgm.scores0, sg_exh_mconf=sgm.mconf)
        t = r_m.Tiler(grid=[2, 3], overlap=15, origin=[0, 0])
        lims, origin = t.compute_limits_by_grid(img0)
        res.update(tiler_limits=np.array([lims[i] for i in sorted(lims)]), tiler_patch=t.extract_patch(img0, lims[4]))
    finally:
        builtins.print = _print
    for k_, v_ in res.items():
        print("   ", k_, None if v_ is None else np.asarray(v_).shape)
    save("g4_wrappers", image0=img0, image1=img1, **res)

    # ---------------- G5: assets pair (config 1), decoded here with PIL ----------------
    from PIL import Image
    a0 = np.asarray(Image.open("/root/reference/assets/img/cam1/IMG_2637.jpg").convert("RGB"))
    a1 = np.asarray(Image.open("/root/reference/assets/img/cam2/IMG_1112.jpg").convert("RGB"))

    def gray(rgb):  # pinned fp32 formula, rounded to u8 (the jpeg never travels)
        f = rgb.astype(np.float32)
        return np.clip(np.rint((0.299 * f[..., 0] + 0.587 * f[..., 1]) + 0.114 * f[..., 2]), 0, 255).astype(np.uint8)

    g0, g1 = gray(a0), gray(a1)
    lg_def = synthetic.lightglue_state_dict(0, "passthrough")
    r_lg.LightGlue.__init__ = _orig_lg_init
    r_sp.SuperPoint.__init__ = _orig_sp_init
    ext = r_sp.SuperPoint(max_num_keypoints=2048).eval()
    ext.load_state_dict(sp_sd)
    mt = r_lg.LightGlue(features="superpoint").eval()
    mt.load_state_dict(lg_def)
    with torch.inference_mode():
        f0 = ext.extract(torch.tensor(g0[None] / 255.0, dtype=torch.float), resize=None)
        f1 = ext.extract(torch.tensor(g1[None] / 255.0, dtype=torch.float), resize=None)
        o = mt({"image0": f0, "image1": f1})
    print(f"  assets: kpts {f0['keypoints'].shape[1]}/{f1['keypoints'].shape[1]} stop={o['stop']} matches={int((o['matches0'] > -1).sum())}")
    gen_colour(sp_sd)
    gen_prune_threshold()
    gen_features_pickle()
    gen_triangulation()
    gen_preselection(sp_sd)
    gen_adaptive_large()
    save("g5_assets", gray0=g0, gray1=g1, keypoints0=f0["keypoints"][0], keypoints1=f1["keypoints"][0],
         scores0=f0["keypoint_scores"][0], scores1=f1["keypoint_scores"][0],
         desc0_sha=sha(f0["descriptors"][0]), desc0_sample=f0["descriptors"][0][::16],
         matches0=o["matches0"][0], matching_scores0=o["matching_scores0"][0], stop=o["stop"],
         prune0=o["prune0"][0], prune1=o["prune1"][0])


if __name__ == "__main__":
    main()
