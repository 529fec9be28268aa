"""The adaptive part of LightGlue - early stop and point pruning (`lightglue/lightglue.py:477-510, 558-585`) - at the sizes of the
headline workload. With trained weights the reference's CPU path prunes after EVERY layer (`:326-331`, quirk q10), so this is the
path a real user takes; the fixtures of rounds 1-4 reached it at <= 300 keypoints only.

  * G9 (tests/golden/g9_lightglue_adaptive_*.npz, written by the REFERENCE's own `LightGlue` in tools/gen_golden.py): synthetic
    features of 2048 / 1536, 4096 / 3000 and 2500 / 4096 points under weights that prune ~30-45 % of the live points per layer
    (`synthetic.lightglue_state_dict(.., "prune_gradual")`) and stop the pair late (`earlystop_late`): device matches, stop layer and
    prune counters (the layer at which each point was dropped, i.e. the live set of every layer) bit-equal, scores within 1e-4;
  * from pixels at 1080p / 4096 keypoints through tests/parity_report.run_case with the same weights: exact or explained, and the
    live counts per layer equal to the oracle's.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from icepy4d_amd import synthetic

pytestmark = pytest.mark.gpu

SP_SD = synthetic.superpoint_state_dict(0)


@pytest.fixture(scope="module")
def big_eng():
    from icepy4d_amd.engine import Engine
    e = Engine(0)
    e.reserve(64, 64, 2, 4096)
    yield e
    e.close()


def run_lightglue(e, f, **conf):
    m, n = f["kpts0"].shape[0], f["kpts1"].shape[0]
    e.kpts.zero_(); e.desc.zero_()
    e.kpts[0, :m] = torch.from_numpy(f["kpts0"]).cuda(); e.kpts[1, :n] = torch.from_numpy(f["kpts1"]).cuda()
    e.desc[0, :m] = torch.from_numpy(f["desc0"]).cuda(); e.desc[1, :n] = torch.from_numpy(f["desc1"]).cuda()
    e.n[:2] = torch.tensor([m, n], dtype=torch.int32)
    e.lightglue(tuple(f["size0"]), tuple(f["size1"]), **conf)
    torch.cuda.synchronize()
    return e.matches_to_host(m, n)


def live_per_layer(prune0, prune1, stop):
    return [[int((prune0 > l).sum()), int((prune1 > l).sum())] for l in range(stop)]


@pytest.mark.parametrize("ci", range(6))
def test_adaptive_path_equals_the_reference_golden(big_eng, ci):
    g = load_golden(f"g9_lightglue_adaptive_{ci}")
    big_eng.load_state_dict("lightglue", synthetic.lightglue_state_dict(0, str(g["variant"])))
    f = synthetic.synthetic_features(int(g["seed"]), int(g["m"]), int(g["n"]))
    out = run_lightglue(big_eng, f, depth_confidence=float(g["depth_confidence"]), width_confidence=float(g["width_confidence"]))
    assert out["stop"] == int(g["stop"])
    assert live_per_layer(out["prune0"], out["prune1"], out["stop"]) == g["live"].tolist()
    assert np.array_equal(out["prune0"], g["prune0"]) and np.array_equal(out["prune1"], g["prune1"])
    assert np.array_equal(out["matches0"], g["matches0"]) and np.array_equal(out["matches1"], g["matches1"])
    assert np.abs(out["matching_scores0"] - g["matching_scores0"]).max() < 1e-4
    assert np.abs(out["matching_scores1"] - g["matching_scores1"]).max() < 1e-4
    if str(g["variant"]) == "prune_gradual":      # the fixture does what it is for: widths in every split-KV regime of its size
        live = g["live"]
        assert (np.diff(live[:, 0]) < 0).all() and live[-1].max() < 0.4 * live[0].max()


@pytest.mark.parametrize("variant,conf", [("prune_gradual", {}), ("prune_gradual", {"depth_confidence": -1}), ("earlystop_late", {})])
def test_adaptive_path_from_pixels_at_1080p_4096(variant, conf):
    """BASELINE configs[1] size from pixels with the adaptive machinery at work in every layer: chain A-D of tests/parity_report.py
    (extraction exact or explained; the oracle's LightGlue on the DEVICE's features gives the device's matches, stop layer and prune
    counters exactly; identical matched pairs from pixels) plus the live counts of every layer equal to the oracle's. The one-channel
    matchability / confidence weights of `prune_gradual` are scaled with the channel statistics of image 0's keypoint descriptors
    (device output; the oracle gets the same weights), which gives the designed walk 4096 -> ~2900 -> ~2200 -> ~1400 -> ~900 -> ~550."""
    import parity_report
    from test_gpu_parity import assert_exact_or_explained
    from icepy4d_amd.engine import Engine
    torch.set_num_threads(min(32, torch.get_num_threads()))
    img0, img1 = synthetic.translated_pair(0, 1080, 1920, 40, 8)
    e = Engine(0)
    e.load_state_dict("superpoint", SP_SD)
    e.reserve(1080, 1920, 2, 4096)
    stats = None
    if variant == "prune_gradual":
        e.superpoint(torch.from_numpy(np.stack([img0, img1])).cuda(), 4, 0.0005, 4, 4096)
        torch.cuda.synchronize()
        d0 = torch.from_numpy(e.features_to_host(0)[1])
        stats = (d0.mean(0), d0.std(0))
    lg_sd = synthetic.lightglue_state_dict(0, variant, channel_stats=stats)
    e.load_state_dict("lightglue", lg_sd)
    rep = parity_report.run_case(e, img0, img1, SP_SD, lg_sd, 4096, lg_conf=conf)
    e.close()
    c = rep["matching_same_features"]
    print(variant, conf, "live", c["live_device"], "stop", c["stop_device"], "matches", c["n_matches_device"], "margins", c["oracle_min_margins"])
    assert_exact_or_explained(rep)
    assert c["live_device"] == c["live_oracle"]
    assert rep["end_to_end"]["pairs_oracle"] > 20
    if variant == "prune_gradual":
        live = np.array(c["live_device"])
        assert any(2048 < w < 4096 for w in live[:, 0]) and any(1024 < w <= 2048 for w in live[:, 0]) and any(w <= 1024 for w in live[:, 0])
        assert c["stop_device"] >= 6
    else:
        assert c["stop_device"] == 7 and all(w == [4096, 4096] for w in c["live_device"])
