"""Decision-margin extractor for the end-to-end parity tests.   *** TEST INFRASTRUCTURE ***

north_star: "keypoints and match indices bit-exact, descriptor scores within 1e-4 fp32". The integer stages of the HIP path
ARE bit-exact on identical inputs (stage-isolated tests). End to end, their inputs are floating-point maps that agree with the
oracle's only to rounding (another accumulation order in the convolutions), so a comparison `a == b` / `a > b` the oracle makes
can come out the other way on the device when |a - b| is below that rounding. This module finds, for every keypoint or match
that differs, the oracle decision that explains it; a difference with no such decision is a parity failure.

Decisions of the extraction stage (`lightglue/superpoint.py:50-72, 177-200`):
  * `simple_nms`: `scores == max_pool(scores)` in three rounds (round 0 on the score map, rounds 1-2 on the map with the
    neighbourhoods of the current maxima zeroed). Margin of pixel p in a round = |v[p] - max(v[window(p) minus p])|. A flipped
    maximum at p changes `supp` within r, `rest` within r, the next round's maxima within 2r, ... : after both recovery rounds
    its influence reaches 4r (Chebyshev), which is the neighbourhood searched.
  * threshold: `scores > detection_threshold`, margin |s - thr|.
  * top-k: margin |s - s_k| to the k-th largest candidate score; and when n candidates above the cut were flipped by one of the
    decisions above, the cut moves by up to n ranks.
Decisions of the assignment stage (`lightglue/lightglue.py:290-306`): row / column arg-max of the log-assignment (gap between
the best and the second best entry) and `mscores > filter_threshold`.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

INF = float("inf")


def _pool(t: torch.Tensor, r: int) -> torch.Tensor:
    return F.max_pool2d(t[None, None], kernel_size=2 * r + 1, stride=1, padding=r)[0, 0]


def _round_margin(v: torch.Tensor, r: int) -> torch.Tensor:
    """|v[p] - max of the other pixels of p's (2r+1)^2 window| for every p (the margin of `v == max_pool(v)` at p)."""
    pooled = _pool(v, r)
    m = pooled - v                       # p is not the maximum: distance to it
    ys, xs = torch.where(m == 0)         # p is a maximum: distance to the runner-up of its window
    if len(ys):
        pad = F.pad(v, (r, r, r, r), value=-INF)
        offs = torch.arange(2 * r + 1)
        for lo in range(0, len(ys), 1 << 16):
            y, x = ys[lo:lo + (1 << 16)], xs[lo:lo + (1 << 16)]
            win = pad[(y[:, None, None] + offs[None, :, None]), (x[:, None, None] + offs[None, None, :])].clone()
            win[:, r, r] = -INF
            m[y, x] = v[y, x] - win.flatten(1).max(1).values
    return m


def nms_margin_map(scores: torch.Tensor, radius: int) -> torch.Tensor:
    """Smallest margin over the three rounds of `simple_nms` at every pixel of one [H, W] score map."""
    zero = torch.zeros_like(scores)
    keep = scores == _pool(scores, radius)
    margin = _round_margin(scores, radius)
    for _ in range(2):
        near = _pool(keep.float(), radius) > 0
        rest = torch.where(near, zero, scores)
        mr = _round_margin(rest, radius)
        mr[near] = INF                   # `(rest == pool(rest)) & ~near_max`: the comparison is masked out there
        margin = torch.minimum(margin, mr)
        keep = keep | ((rest == _pool(rest, radius)) & ~near)
    return margin


def explain_keypoint_diffs(score_map: torch.Tensor, nms_map: torch.Tensor, kp_ours: np.ndarray, kp_ref: np.ndarray,
                           radius: int, border: int, threshold: float, max_k: Optional[int], eps: float) -> Dict:
    """score_map / nms_map: the ORACLE's [H, W] maps. kp_*: [n, 2] (x, y). Returns the keypoints of the symmetric difference
    with the oracle decision (margin <= eps) that explains each, and the list of unexplained ones."""
    ours = {(int(x), int(y)) for x, y in kp_ours}
    ref = {(int(x), int(y)) for x, y in kp_ref}
    diff = sorted(ours ^ ref)
    out = {"n_ours": len(ours), "n_ref": len(ref), "n_diff": len(diff), "reasons": {}, "unexplained": []}
    if not diff:
        return out
    h, w = score_map.shape
    reach = 4 * radius
    mm = nms_margin_map(score_map, radius)
    near_tie = -_pool(-mm, reach)        # smallest margin within Chebyshev distance 4r
    s = nms_map.clone()
    if border:
        s[:border] = -1; s[:, :border] = -1; s[-border:] = -1; s[:, -border:] = -1
    cand = torch.sort(s[s > threshold], descending=True).values
    cut_active = max_k is not None and 0 <= max_k < len(cand)
    kth = float(cand[max_k - 1]) if cut_active and max_k > 0 else None
    n_flipped_above = 0
    pending = []
    for (x, y) in diff:
        sv = float(score_map[y, x])
        why = []
        if float(near_tie[y, x]) <= eps:
            why.append("nms")
        if abs(sv - threshold) <= eps:
            why.append("threshold")
        if cut_active and abs(sv - kth) <= eps:
            why.append("cut")
        if why:
            if cut_active and sv > kth:
                n_flipped_above += 1
            for k in why:
                out["reasons"][k] = out["reasons"].get(k, 0) + 1
        else:
            pending.append((x, y, sv))
    for (x, y, sv) in pending:
        # the cut moved: a candidate within n_flipped_above ranks of the k-th is pushed across it
        ok = False
        if cut_active and n_flipped_above:
            rank = int((cand > sv).sum())
            ok = abs(rank - max_k) <= n_flipped_above and float(nms_map[y, x]) > 0
        if ok:
            out["reasons"]["cut-shift"] = out["reasons"].get("cut-shift", 0) + 1
        else:
            out["unexplained"].append((x, y, sv, float(near_tie[y, x])))
    return out


def explain_order_diffs(kp_ours: np.ndarray, kp_ref: np.ndarray, sc_ref: np.ndarray, eps: float) -> Dict:
    """Same keypoint SET on both sides but listed in another order (`torch.topk(sorted=True)` orders by score): a keypoint
    found at rank i on the device and at rank j in the oracle is explained when the oracle scores at ranks i and j differ by
    <= eps (equal scores: torch.topk's order among ties is unspecified)."""
    pos = {(int(x), int(y)): j for j, (x, y) in enumerate(kp_ref)}
    moved = [i for i in range(min(len(kp_ours), len(kp_ref))) if tuple(kp_ours[i]) != tuple(kp_ref[i])]
    out = {"n_moved": len(moved), "n_exact_ties": 0, "max_gap": 0.0, "unexplained": []}
    for i in moved:
        j = pos.get((int(kp_ours[i][0]), int(kp_ours[i][1])))
        if j is None:
            continue                     # not in the oracle's set: reported by explain_keypoint_diffs
        gap = abs(float(sc_ref[i]) - float(sc_ref[j]))
        out["max_gap"] = max(out["max_gap"], gap)
        if gap == 0.0:
            out["n_exact_ties"] += 1
        elif gap > eps:
            out["unexplained"].append((i, j, gap))
    return out


def explain_match_diffs(log_assign: torch.Tensor, m0_ours: np.ndarray, m0_ref: np.ndarray, filter_threshold: float,
                        eps: float) -> Dict:
    """log_assign: the ORACLE's [M+1, N+1] log-assignment (or optimal-transport) matrix for the SAME features both sides
    matched. A differing matches0[i] is explained by an arg-max gap (row i, or the column of either partner) or by the
    threshold `exp(score) > filter_threshold` within eps."""
    inner = log_assign[:-1, :-1]
    idx = np.where(np.asarray(m0_ours) != np.asarray(m0_ref))[0]
    out = {"n": int(len(m0_ref)), "n_diff": int(len(idx)), "reasons": {}, "unexplained": []}
    if not len(idx):
        return out

    def gap(vec):
        if vec.numel() < 2:
            return INF
        t = torch.topk(vec, 2).values
        return float(t[0] - t[1])

    for i in idx:
        why = []
        row = inner[i]
        if gap(row) <= eps:
            why.append("row-argmax")
        best = float(row.max())
        if abs(float(np.exp(best)) - filter_threshold) <= eps:
            why.append("threshold")
        for j in {int(m0_ours[i]), int(m0_ref[i]), int(row.argmax())}:
            if j >= 0 and gap(inner[:, j]) <= eps:
                why.append("col-argmax")
                break
        if why:
            for k in why:
                out["reasons"][k] = out["reasons"].get(k, 0) + 1
        else:
            out["unexplained"].append((int(i), int(m0_ours[i]), int(m0_ref[i]), gap(row)))
    return out


def match_pairs(kp0: np.ndarray, kp1: np.ndarray, m0: np.ndarray) -> set:
    """Matches as coordinate pairs ((x0, y0), (x1, y1)): independent of the order keypoints are listed in."""
    return {(tuple(kp0[i]), tuple(kp1[j])) for i, j in enumerate(m0) if j > -1}


def assert_same_matches(k0, k1, m0, ref_k0, ref_k1, ref_m0, ref_s0=None, ref_s1=None):
    """Match indices bit-exact when both sides list the keypoints in the same order; when the top-k order differs among
    scores closer than the float error (tests/margins.py), the same matched COORDINATE pairs."""
    if np.array_equal(k0, ref_k0) and np.array_equal(k1, ref_k1):
        assert np.array_equal(m0, ref_m0), int(np.sum(m0 != ref_m0))
        return
    assert {tuple(p) for p in k0} == {tuple(p) for p in ref_k0} and {tuple(p) for p in k1} == {tuple(p) for p in ref_k1}
    if ref_s0 is not None:
        assert explain_order_diffs(k0, ref_k0, ref_s0, 1e-6)["unexplained"] == []
        assert explain_order_diffs(k1, ref_k1, ref_s1, 1e-6)["unexplained"] == []
    assert match_pairs(k0, k1, m0) == match_pairs(ref_k0, ref_k1, ref_m0)
