"""End-to-end parity report of the HIP path against the oracle: mismatch COUNTS per case, each mismatch traced to the oracle
decision margin that explains it (tests/margins.py).   *** TEST INFRASTRUCTURE (imports oracle/) ***

    python tests/parity_report.py --out gpurun_out/parity_wino.json            # Winograd convolutions (the default path)
    IM_CONV_DIRECT=1 python tests/parity_report.py --out gpurun_out/parity_direct.json

`run_case` is also what tests/test_gpu_parity.py asserts on (zero unexplained mismatches).

The chain that is verified, per case:
  A. extraction: device score map vs oracle score map (max abs error, asserted < 5e-7 by the tests; the float tolerance eps_s := 4 x that, capped at 2e-6);
     device keypoints vs oracle keypoints: every keypoint of the symmetric difference must be explained by an oracle decision
     (NMS equality / threshold / top-k cut) with margin <= eps_s; keypoints present on both sides carry scores within 1e-5 and
     descriptors within 1e-4;
  B. extraction decisions alone: the oracle's integer stages (simple_nms, border/threshold/top-k) run on the DEVICE's score
     map must reproduce the device keypoints exactly (ties as sets);
  C. matching: the oracle's LightGlue run on the DEVICE's features (identical inputs) must give the device's matches0 / stop /
     prune exactly, or each differing index explained by an arg-max gap / threshold margin <= 1e-4;
  D. end to end (device vs oracle from pixels): identical when A found no difference and C none; otherwise reported as
     coordinate-pair overlap (a flipped keypoint changes every descriptor through attention, so no margin applies).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import margins  # noqa: E402
from icepy4d_amd import synthetic  # noqa: E402


# The score map of the device differs from the oracle's by another fp32 summation order only: 0.7-2.1e-7 over every case of
# rounds 1-3 (profiles/r0*_parity_*.json). The tests assert err < ERR_BOUND, and an oracle decision margin may "explain" a
# flipped keypoint only below eps = min(4 err, EPS_CAP): a convolution that drifted to 2e-6 fails instead of explaining more.
ERR_BOUND = 5e-7
EPS_CAP = 2e-6


def _read(eng, name, numel):
    from icepy4d_amd._lib import stream_ptr
    buf = torch.empty(numel, device="cuda")
    eng.ctx.call("im_debug_read", name.encode(), buf.data_ptr(), numel, stream_ptr())
    return buf.cpu()


def _tie_groups_equal(kp, sc, ref_kp, ref_sc):
    if kp.shape != ref_kp.shape or not np.array_equal(sc, ref_sc):
        return False
    if np.array_equal(kp, ref_kp):
        return True
    for v in np.unique(sc[np.any(kp != ref_kp, axis=1)]):
        idx = np.where(sc == v)[0]
        if {tuple(p) for p in kp[idx]} != {tuple(p) for p in ref_kp[idx]}:
            return False
    return True


def run_case(eng, img0, img1, sp_sd, lg_sd, max_k, lg_conf=None, radius=4, thr=0.0005, border=4, match=True):
    """One stereo pair through the device path and the oracle; returns the report dict (see module docstring)."""
    from oracle import ref_cpu as o
    lg_conf = lg_conf or {}
    h, w = img0.shape
    assert img1.shape == img0.shape
    H8, W8 = (h // 8) * 8, (w // 8) * 8
    eng.reserve(h, w, 2, max_k)
    K = eng.max_kpts
    pair = torch.from_numpy(np.stack([img0, img1])).cuda()
    eng.superpoint(pair, radius, thr, border, max_k)
    torch.cuda.synchronize()
    smap_d = _read(eng, "sp_smap", 2 * H8 * W8).view(2, H8, W8)
    rep = {"shape": [h, w], "max_k": max_k, "images": []}
    feats_d, feats_o = [], []
    eps_all = 0.0
    for b, img in enumerate((img0, img1)):
        kp, desc, sc = eng.features_to_host(b)
        with torch.inference_mode():
            tr = {}
            ref = o.superpoint_lg(o.frame_to_tensor(img), sp_sd, max_k, radius, thr, border, trace=tr)
        smap_o, nms_o = tr["score_map"][0], tr["nms"][0]
        err = float((smap_d[b] - smap_o).abs().max())
        eps = min(max(4.0 * err, 1e-9), EPS_CAP)
        eps_all = max(eps_all, eps)
        ref_kp, ref_sc, ref_desc = ref["keypoints"].numpy(), ref["keypoint_scores"].numpy(), ref["descriptors"].numpy()
        ex = margins.explain_keypoint_diffs(smap_o, nms_o, kp, ref_kp, radius, border, thr, max_k, eps)
        ours = {tuple(p): i for i, p in enumerate(kp)}
        common = [(ours[tuple(p)], j) for j, p in enumerate(ref_kp) if tuple(p) in ours]
        ii, jj = (np.array(common).T if common else (np.zeros(0, int), np.zeros(0, int)))
        # B: the oracle's integer stages on the device's score map
        with torch.inference_mode():
            nms_on_d = o.simple_nms(smap_d[b][None], radius)[0]
            kp_b, sc_b = o.select_keypoints_lg(nms_on_d, border, thr, max_k)
        stage_exact = _tie_groups_equal(kp, sc, kp_b.numpy(), sc_b.numpy())
        order = margins.explain_order_diffs(kp, ref_kp, ref_sc, eps)
        rep["images"].append({
            "score_map_max_abs_err": err, "eps": eps, "n_keypoints": int(len(kp)), "n_keypoints_oracle": int(len(ref_kp)),
            "keypoint_set_diff": ex["n_diff"], "diff_reasons": ex["reasons"], "unexplained": ex["unexplained"],
            "same_order": bool(kp.shape == ref_kp.shape and np.array_equal(kp, ref_kp)),
            "ranks_moved": order["n_moved"], "ranks_moved_exact_ties": order["n_exact_ties"], "ranks_moved_max_score_gap": order["max_gap"],
            "ranks_moved_unexplained": order["unexplained"],
            "score_max_abs_err_common": float(np.abs(sc[ii] - ref_sc[jj]).max()) if len(ii) else 0.0,
            "desc_max_abs_err_common": float(np.abs(desc[ii] - ref_desc[jj]).max()) if len(ii) else 0.0,
            "integer_stages_exact_on_device_map": bool(stage_exact)})
        feats_d.append((kp, desc, sc))
        feats_o.append(ref)
    if not match:
        return rep
    eng.lightglue((w, h), (w, h), **lg_conf)
    torch.cuda.synchronize()
    (k0, d0, s0), (k1, d1, s1) = feats_d
    out = eng.matches_to_host(len(k0), len(k1))
    size = torch.tensor([w, h], dtype=torch.float)
    with torch.inference_mode():
        tr = {}
        f0 = dict(keypoints=torch.from_numpy(k0), descriptors=torch.from_numpy(d0), image_size=size)
        f1 = dict(keypoints=torch.from_numpy(k1), descriptors=torch.from_numpy(d1), image_size=size)
        same = o.lightglue(f0, f1, lg_sd, trace=tr, **lg_conf)          # C: oracle matcher on the device's features
        e2e = o.lightglue(feats_o[0], feats_o[1], lg_sd, **lg_conf)     # D: oracle from pixels
    m0_same = same["matches0"].numpy()
    pruned = len(tr["kept0"]) != len(k0) or len(tr["kept1"]) != len(k1)
    if pruned:   # log_assignment is in the compact index space: compare in that space
        kept0, kept1 = tr["kept0"].numpy(), tr["kept1"].numpy()
        inv1 = -np.ones(len(k1) + 1, np.int64); inv1[kept1] = np.arange(len(kept1))
        mo = inv1[out["matches0"][kept0]]; mr = inv1[m0_same[kept0]]
        exm = margins.explain_match_diffs(tr["log_assignment"], mo, mr, float(lg_conf.get("filter_threshold", 0.1)), 1e-4)
        outside = int(np.sum(np.delete(out["matches0"], kept0) != -1))
        exm["matched_but_pruned_in_oracle"] = outside
    else:
        exm = margins.explain_match_diffs(tr["log_assignment"], out["matches0"], m0_same,
                                          float(lg_conf.get("filter_threshold", 0.1)), 1e-4)
    v = (out["matches0"] > -1) & (m0_same > -1)
    rep["matching_same_features"] = {
        "n0": int(len(k0)), "n1": int(len(k1)), "n_matches_device": int((out["matches0"] > -1).sum()),
        "n_matches_oracle": int((m0_same > -1).sum()), "matches0_diff": exm["n_diff"], "diff_reasons": exm["reasons"],
        "unexplained": exm["unexplained"], "stop_device": out["stop"], "stop_oracle": int(same["stop"]),
        "prune0_equal": bool(np.array_equal(out["prune0"], same["prune0"].numpy())),
        "prune1_equal": bool(np.array_equal(out["prune1"], same["prune1"].numpy())),
        # live points per layer (prune counter = 1 + the number of layers a point survived, `lightglue.py:481-482, 502, 510`)
        "live_device": [[int((out["prune0"] > l).sum()), int((out["prune1"] > l).sum())] for l in range(out["stop"])]
        if float(lg_conf.get("width_confidence", 0.99)) > 0 else [],
        "live_oracle": [[len(l["ind0"]), len(l["ind1"])] for l in tr["layers"]],
        "oracle_min_margins": {k: min((l[k] for l in tr["layers"] if k in l), default=None)
                               for k in ("token_margin", "stop_margin", "match_margin0", "match_margin1")},
        "mscore_max_abs_err": float(np.abs(out["matching_scores0"][v] - same["matching_scores0"].numpy()[v]).max()) if v.any() else 0.0}
    ours = margins.match_pairs(k0, k1, out["matches0"])
    theirs = margins.match_pairs(feats_o[0]["keypoints"].numpy(), feats_o[1]["keypoints"].numpy(), e2e["matches0"].numpy())
    rep["end_to_end"] = {"pairs_device": len(ours), "pairs_oracle": len(theirs), "pairs_common": len(ours & theirs),
                         "stop_device": out["stop"], "stop_oracle": int(e2e["stop"]),
                         "identical": bool(ours == theirs and out["stop"] == int(e2e["stop"]))}
    return rep


def run_case_superglue(eng, img0, img1, sp_sd, sg_sd, max_k, radius=3, thr=0.001, border=4, iters=20, match_thr=0.3, check_ot=False):
    """The same chain A-D for the SuperGlue flavour (`SuperGlueMatcher._match_images`, `matchers.py:892-940`): MagicLeap-flavour
    SuperPoint selection, keypoint encoder + 18 GNN layers + optimal transport + mutual filter; C uses the oracle's optimal
    transport matrix of the DEVICE's features for the arg-max / threshold margins. The engine must hold `sg_sd`.
    check_ot: additionally (E) the device's score matrix (`superglue.py:279-280`) against the oracle's on the same features, and the
    device's Sinkhorn (`im_log_optimal_transport`) against `o.log_optimal_transport` on the DEVICE's OWN score matrix
    (`superglue.py:152-186`): max abs differences, at the full (n0 + 1) x (n1 + 1) size."""
    from oracle import ref_cpu as o
    h, w = img0.shape
    assert img1.shape == img0.shape
    H8, W8 = (h // 8) * 8, (w // 8) * 8
    eng.reserve(h, w, 2, max_k)
    pair = torch.from_numpy(np.stack([img0, img1])).cuda()
    eng.superpoint(pair, radius, thr, border, max_k, flavour=1)
    torch.cuda.synchronize()
    smap_d = _read(eng, "sp_smap", 2 * H8 * W8).view(2, H8, W8)
    rep = {"shape": [h, w], "max_k": max_k, "flavour": "superglue", "images": []}
    feats_d, feats_o = [], []
    for b, img in enumerate((img0, img1)):
        kp, desc, sc = eng.features_to_host(b)
        with torch.inference_mode():
            feat = o.sp_encoder(torch.tensor(img / 255.0, dtype=torch.float)[None, None], sp_sd)   # `matchers.py:263-274`
            smap_o = o.sp_score_map(feat, sp_sd)
            nms_o = o.simple_nms(smap_o, radius)
            ref_kp_t, ref_sc_t = o.select_keypoints_sg(nms_o[0], border, thr, max_k)
            ref_desc_t = o.sample_descriptors(ref_kp_t, o.sp_dense_descriptors(feat, sp_sd)[0])     # [256, K]
            nms_on_d = o.simple_nms(smap_d[b][None], radius)[0]
            kp_b, sc_b = o.select_keypoints_sg(nms_on_d, border, thr, max_k)
        err = float((smap_d[b] - smap_o[0]).abs().max())
        eps = min(max(4.0 * err, 1e-9), EPS_CAP)
        ref_kp, ref_sc, ref_desc = ref_kp_t.numpy(), ref_sc_t.numpy(), ref_desc_t.numpy().T
        ex = margins.explain_keypoint_diffs(smap_o[0], nms_o[0], kp, ref_kp, radius, border, thr, max_k, eps)
        ours = {tuple(q): i for i, q in enumerate(kp)}
        common = [(ours[tuple(q)], j) for j, q in enumerate(ref_kp) if tuple(q) in ours]
        ii, jj = (np.array(common).T if common else (np.zeros(0, int), np.zeros(0, int)))
        order = margins.explain_order_diffs(kp, ref_kp, ref_sc, eps)
        rep["images"].append({
            "score_map_max_abs_err": err, "eps": eps, "n_keypoints": int(len(kp)), "n_keypoints_oracle": int(len(ref_kp)),
            "keypoint_set_diff": ex["n_diff"], "diff_reasons": ex["reasons"], "unexplained": ex["unexplained"],
            "same_order": bool(kp.shape == ref_kp.shape and np.array_equal(kp, ref_kp)),
            "ranks_moved": order["n_moved"], "ranks_moved_exact_ties": order["n_exact_ties"], "ranks_moved_max_score_gap": order["max_gap"],
            "ranks_moved_unexplained": order["unexplained"],
            "score_max_abs_err_common": float(np.abs(sc[ii] - ref_sc[jj]).max()) if len(ii) else 0.0,
            "desc_max_abs_err_common": float(np.abs(desc[ii] - ref_desc[jj]).max()) if len(ii) else 0.0,
            "integer_stages_exact_on_device_map": bool(_tie_groups_equal(kp, sc, kp_b.numpy(), sc_b.numpy()))})
        feats_d.append((kp, desc, sc))
        feats_o.append((ref_kp_t, ref_desc_t, ref_sc_t))
    eng.superglue((h, w), (h, w), sinkhorn_iterations=iters, match_threshold=match_thr)
    torch.cuda.synchronize()
    (k0, d0, s0), (k1, d1, s1) = feats_d
    out = eng.matches_to_host(len(k0), len(k1))

    def data(f0, f1):
        return dict(keypoints0=f0[0], keypoints1=f1[0], descriptors0=f0[1], descriptors1=f1[1], scores0=f0[2], scores1=f1[2],
                    shape0=(h, w), shape1=(h, w))
    with torch.inference_mode():
        tr = {}
        dev0 = (torch.from_numpy(k0), torch.from_numpy(d0).T.contiguous(), torch.from_numpy(s0))
        dev1 = (torch.from_numpy(k1), torch.from_numpy(d1).T.contiguous(), torch.from_numpy(s1))
        same = o.superglue(data(dev0, dev1), sg_sd, iters, match_thr, trace=tr)      # C: oracle matcher on the device's features
        e2e = o.superglue(data(feats_o[0], feats_o[1]), sg_sd, iters, match_thr)     # D: oracle from pixels
    m0_same = same["matches0"].numpy()
    exm = margins.explain_match_diffs(tr["ot"], out["matches0"], m0_same, match_thr, 1e-4)
    v = (out["matches0"] > -1) & (m0_same > -1)
    rep["matching_same_features"] = {
        "n0": int(len(k0)), "n1": int(len(k1)), "n_matches_device": int((out["matches0"] > -1).sum()),
        "n_matches_oracle": int((m0_same > -1).sum()), "matches0_diff": exm["n_diff"], "diff_reasons": exm["reasons"],
        "unexplained": exm["unexplained"], "stop_device": 0, "stop_oracle": 0, "prune0_equal": True, "prune1_equal": True,
        "mscore_max_abs_err": float(np.abs(out["matching_scores0"][v] - same["matching_scores0"].numpy()[v]).max()) if v.any() else 0.0}
    if check_ot:
        from icepy4d_amd._lib import ptr
        K, n0, n1 = eng.max_kpts, len(k0), len(k1)
        sim_d = torch.empty(K * K, device=eng.device)
        eng.ctx.call("im_debug_read", b"sim", sim_d.data_ptr(), K * K, eng.stream_ptr())      # the forward's own score matrix, row stride K
        zout = torch.empty((n0 + 1) * (n1 + 1), device=eng.device)
        eng.ctx.call("im_log_optimal_transport", ptr(sim_d), n0, n1, K, float(sg_sd["bin_score"]), iters, ptr(zout), eng.stream_ptr())
        torch.cuda.synchronize()
        sim_h = sim_d.view(K, K)[:n0, :n1].cpu().contiguous()
        z_d = zout.view(n0 + 1, n1 + 1).cpu()
        del sim_d, zout
        with torch.inference_mode():
            z_o = o.log_optimal_transport(sim_h[None], sg_sd["bin_score"], iters)[0]
        rep["sinkhorn_on_device_scores"] = {
            "rows": n0 + 1, "cols": n1 + 1, "iterations": iters,
            "scores_max_abs_err_vs_oracle_same_features": float((sim_h - tr["scores"]).abs().max()),
            "ot_max_abs_err": float((z_d - z_o).abs().max()),
            "ot_max_abs_err_vs_oracle_same_features": float((z_d - tr["ot"]).abs().max())}
        del z_o, z_d, sim_h
    ours = margins.match_pairs(k0, k1, out["matches0"])
    theirs = margins.match_pairs(feats_o[0][0].numpy(), feats_o[1][0].numpy(), e2e["matches0"].numpy())
    rep["end_to_end"] = {"pairs_device": len(ours), "pairs_oracle": len(theirs), "pairs_common": len(ours & theirs),
                         "identical": bool(ours == theirs)}
    return rep


def cases(full: bool):
    from conftest import load_golden
    g1a, g1b, g4, g5 = (load_golden(n) for n in ("g1_superpoint_a", "g1_superpoint_b", "g4_wrappers", "g5_assets"))
    yield "g1_a (96x128, K=64)", g1a["image"], g1a["image"], 64
    yield "g1_b (136x200, K=512)", g1b["image"], g1b["image"], 512
    yield "g4 wrappers pair (200x304, K=256)", g4["image0"], g4["image1"], 256
    yield "g5 assets pair = config 1 (800x1200, K=2048)", g5["gray0"], g5["gray1"], 2048
    if full:
        a, b = synthetic.translated_pair(0, 1080, 1920, 40, 8)
        yield "config 2 translated pair (1080x1920, K=4096)", a, b, 4096
        a, b = synthetic.stereo_pair(0, 1080, 1920)
        yield "config 2 bench pair, epoch 0 (1080x1920, K=4096)", a, b, 4096


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity.json"))
    ap.add_argument("--no-full", action="store_true")
    ap.add_argument("--epochs", type=int, default=0, help="additional configs[1] bench pairs (epochs 1..N), LightGlue")
    ap.add_argument("--superglue", action="store_true", help="SuperGlue-flavour cases (SuperPoint nms 3 + SuperGlue), incl. 1080p / 4096")
    ap.add_argument("--adaptive", type=int, default=0, help="ONLY the adaptive-path campaign: bench epochs 0..N-1 (1080x1920, K=4096) under the "
                    "`prune_gradual` weights (default options and depth_confidence=-1) and `earlystop_late`: pruning / early stop at work in every layer, "
                    "live counts per layer against the oracle (tests/test_gpu_adaptive.py runs one pair of each)")
    ap.add_argument("--config5", action="store_true", help="ONLY the BASELINE configs[4] size: 3000x4000 pair, 16384 keypoints, SuperGlue with 20 "
                    "Sinkhorn iterations, incl. the Sinkhorn / score-matrix comparison at 16385 x 16385 (minutes of oracle time)")
    ap.add_argument("--config5-mid", action="store_true", help="with --config5: also the 2000x3000 / 8192-keypoint case of the GPU suite")
    args = ap.parse_args()
    from icepy4d_amd.engine import Engine
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    sp_sd = synthetic.superpoint_state_dict(0)
    lg_sd = synthetic.lightglue_state_dict(0, "passthrough")
    eng = Engine(0)
    eng.load_state_dict("superpoint", sp_sd)
    eng.load_state_dict("lightglue", lg_sd)
    report = {"conv": "direct" if os.environ.get("IM_CONV_DIRECT") == "1" else "winograd", "cases": {}}
    if args.adaptive:
        # calibration of the one-channel weights: keypoint descriptors of image 0 of epoch 0, from the device (as in tests/test_gpu_adaptive.py)
        a, b = synthetic.stereo_pair(0, 1080, 1920)
        eng.reserve(1080, 1920, 2, 4096)
        eng.superpoint(torch.from_numpy(np.stack([a, b])).cuda(), 4, 0.0005, 4, 4096)
        torch.cuda.synchronize()
        d0 = torch.from_numpy(eng.features_to_host(0)[1])
        stats = (d0.mean(0), d0.std(0))
        live_equal = 0
        for variant, conf in (("prune_gradual", {}), ("prune_gradual", {"depth_confidence": -1}), ("earlystop_late", {})):
            v_sd = synthetic.lightglue_state_dict(0, variant, channel_stats=stats if variant == "prune_gradual" else None)
            eng.load_state_dict("lightglue", v_sd)
            for e in range(args.adaptive):
                a, b = (synthetic.translated_pair(e, 1080, 1920, 40, 8) if e % 2 else synthetic.stereo_pair(e, 1080, 1920))
                name = f"adaptive {variant} {conf or 'default options'}: {'translated' if e % 2 else 'bench'} pair, epoch {e} (1080x1920, K=4096)"
                t = time.time()
                report["cases"][name] = run_case(eng, a, b, sp_sd, v_sd, 4096, lg_conf=conf)
                c = report["cases"][name]["matching_same_features"]
                live_equal += int(c["live_device"] == c["live_oracle"] and c["stop_device"] == c["stop_oracle"] and c["prune0_equal"] and c["prune1_equal"])
                report["cases"][name]["seconds"] = round(time.time() - t, 1)
                print(name, json.dumps(report["cases"][name]), flush=True)
        report["adaptive_cases_with_equal_stop_layer_prune_counters_and_live_counts"] = live_equal
    if args.config5:
        sg_sd = synthetic.superglue_state_dict(0, "passthrough")
        eng.load_state_dict("superglue", sg_sd)
        big = []
        if args.config5_mid:
            big.append(("superglue: translated pair (2000x3000, K=8192)",) + synthetic.translated_pair(6, 2000, 3000, 48, 16) + (8192,))
        big.append(("superglue: configs[4] translated pair (3000x4000, K=16384)",) + synthetic.translated_pair(5, 3000, 4000, 48, 16) + (16384,))
        for name, a, b, k in big:
            t = time.time()
            report["cases"][name] = run_case_superglue(eng, a, b, sp_sd, sg_sd, k, check_ot=True)
            report["cases"][name]["seconds"] = round(time.time() - t, 1)
            print(name, json.dumps(report["cases"][name]), flush=True)
    for name, a, b, k in ([] if (args.config5 or args.adaptive) else cases(not args.no_full)):
        t = time.time()
        report["cases"][name] = run_case(eng, a, b, sp_sd, lg_sd, k)
        report["cases"][name]["seconds"] = round(time.time() - t, 1)
        print(name, json.dumps(report["cases"][name]), flush=True)
    for e in range(1, args.epochs + 1):
        a, b = synthetic.stereo_pair(e, 1080, 1920)
        name = f"config 2 bench pair, epoch {e} (1080x1920, K=4096)"
        report["cases"][name] = run_case(eng, a, b, sp_sd, lg_sd, 4096)
        print(name, json.dumps(report["cases"][name]), flush=True)
    if args.superglue:
        from conftest import load_golden
        sg_sd = synthetic.superglue_state_dict(0, "passthrough")
        eng.load_state_dict("superglue", sg_sd)
        g4, g5 = load_golden("g4_wrappers"), load_golden("g5_assets")
        sg_cases = [("superglue: g4 wrappers pair (200x304, K=256)", g4["image0"], g4["image1"], 256),
                    ("superglue: g5 assets pair (800x1200, K=2048)", g5["gray0"], g5["gray1"], 2048)]
        if not args.no_full:
            a, b = synthetic.translated_pair(0, 1080, 1920, 40, 8)
            sg_cases.append(("superglue: translated pair (1080x1920, K=4096)", a, b, 4096))
        for name, a, b, k in sg_cases:
            t = time.time()
            report["cases"][name] = run_case_superglue(eng, a, b, sp_sd, sg_sd, k)
            report["cases"][name]["seconds"] = round(time.time() - t, 1)
            print(name, json.dumps(report["cases"][name]), flush=True)
    tot = {"cases": len(report["cases"]), "images": 0, "keypoints": 0, "keypoint_set_diff": 0, "keypoint_unexplained": 0, "ranks_moved": 0,
           "ranks_unexplained": 0, "matches0_compared": 0, "matches0_diff": 0, "matches0_unexplained": 0, "end_to_end_identical": 0}
    for c in report["cases"].values():
        for im in c["images"]:
            tot["images"] += 1
            tot["keypoints"] += im["n_keypoints"]
            tot["keypoint_set_diff"] += im["keypoint_set_diff"]
            tot["keypoint_unexplained"] += len(im["unexplained"])
            tot["ranks_moved"] += im["ranks_moved"]
            tot["ranks_unexplained"] += len(im["ranks_moved_unexplained"])
        m = c["matching_same_features"]
        tot["matches0_compared"] += m["n0"]
        tot["matches0_diff"] += m["matches0_diff"]
        tot["matches0_unexplained"] += len(m["unexplained"])
        tot["end_to_end_identical"] += int(c["end_to_end"]["identical"])
    report["totals"] = tot
    print("totals", json.dumps(tot), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(report, fh, indent=1)
    eng.close()


if __name__ == "__main__":
    main()
