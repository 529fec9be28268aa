"""Geometric verification of matches by a fundamental-matrix RANSAC (row f-2 of the scope table: the step right
after the hot path). The reference delegates to pydegensac, falling back to OpenCV USAC_MAGSAC and finally to
"all inliers" (`src/icepy4d/matching/geometric_verification.py:55-100`); both are un-vendored C++ libraries that are
absent here, so the two enum values are restatements of the published algorithms those calls run (parity with the
libraries' randomised outputs is statistical only and unpinned):

* PYDEGENSAC = LO-RANSAC with a degeneracy check (Chum et al., DEGENSAC): normalised 8-point hypotheses scored by Sampson
  error against `threshold`; the number of hypotheses follows `confidence` (log(1 - conf) / log(1 - w^8), capped by
  `max_iters`); local optimisation = re-fit on the inliers and re-score until the inlier count stops growing; degeneracy =
  if the inliers are dominated by one plane (a homography explains most of them) the plane-and-parallax model F = [e']x H
  with the epipole from two off-plane matches is tried and kept when it has more inliers.
* MAGSAC = what the reference's fallback call runs, `cv2.findFundamentalMat(.., USAC_MAGSAC, 0.5, 0.999, 100000)`: it
  IGNORES the caller's threshold / confidence (`geometric_verification.py:89-91`), and so does this: sigma_max = 0.5 px, the
  best hypothesis is refined by iteratively re-weighted least squares with the MAGSAC++ weights (residuals marginalised over
  the noise scale, chi distribution with 4 degrees of freedom, 0.99 quantile k = 3.64), inliers = residual <= k sigma_max.

Hypotheses are generated and scored in one launch pair on the device when `engine=...` is given (`im_ransac_fundamental`,
csrc/geometry.hip: what `match()` uses), in batches until the confidence criterion is met; without an engine the numpy loop
below runs (the checker of the device form in the tests, and what a host-only caller gets)."""
import logging
from typing import Tuple

import numpy as np

from .enums import GeometricVerification

logger = logging.getLogger(__name__)


def _normalise(p: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    c = p.mean(0)
    d = np.sqrt(((p - c) ** 2).sum(1)).mean()
    s = np.sqrt(2.0) / max(d, 1e-12)
    T = np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])
    return np.c_[p, np.ones(len(p))] @ T.T, T


def _eight_point(p0: np.ndarray, p1: np.ndarray) -> np.ndarray:
    x0, T0 = _normalise(p0)
    x1, T1 = _normalise(p1)
    A = np.einsum("ni,nj->nij", x1, x0).reshape(len(p0), 9)
    _, _, vt = np.linalg.svd(A, full_matrices=len(p0) < 9)   # 8 x 9 needs the full V^T for its null vector; the refit does not
    F = vt[-1].reshape(3, 3)
    u, s, vt = np.linalg.svd(F)
    F = u @ np.diag([s[0], s[1], 0.0]) @ vt
    F = T1.T @ F @ T0
    return F / max(np.linalg.norm(F), 1e-12)


def _sampson(F: np.ndarray, p0: np.ndarray, p1: np.ndarray) -> np.ndarray:
    x0 = np.c_[p0, np.ones(len(p0))]
    x1 = np.c_[p1, np.ones(len(p1))]
    Fx0 = x0 @ F.T
    Ftx1 = x1 @ F
    num = (x1 * Fx0).sum(1) ** 2
    den = Fx0[:, 0] ** 2 + Fx0[:, 1] ** 2 + Ftx1[:, 0] ** 2 + Ftx1[:, 1] ** 2
    return num / np.maximum(den, 1e-24)


MAGSAC_SIGMA_MAX, MAGSAC_K, MAGSAC_CONF, MAGSAC_ITERS = 0.5, 3.64, 0.999, 100000   # `geometric_verification.py:89-91`
DEVICE_BATCH = 1024                                                               # hypotheses per launch pair


def _needed(confidence: float, w: float) -> int:
    w = min(max(w, 1e-9), 1 - 1e-9)
    return int(np.ceil(np.log(max(1 - confidence, 1e-12)) / np.log(1 - w ** 8)))


def geometric_verification(mkpts0: np.ndarray = None, mkpts1: np.ndarray = None,
                           method: GeometricVerification = GeometricVerification.PYDEGENSAC, threshold: float = 1,
                           confidence: float = 0.9999, max_iters: int = 10000, seed: int = 0, engine=None,
                           enable_degeneracy_check: bool = True, **_ignored):
    """Returns (F [3,3] or None, inlier mask [S] bool), like the reference (`geometric_verification.py:11-102`)."""
    assert isinstance(method, GeometricVerification), "Invalid method. It must be a GeometricVerification enum"
    n = 0 if mkpts0 is None else len(mkpts0)
    if method == GeometricVerification.NONE or n < 8:
        if n < 8 and method != GeometricVerification.NONE:
            logger.warning("Not enough matches for geometric verification: all matches kept")
        return None, np.ones(n, dtype=bool)
    p0, p1 = np.asarray(mkpts0, np.float64), np.asarray(mkpts1, np.float64)
    magsac = method == GeometricVerification.MAGSAC
    if magsac:
        threshold, confidence, max_iters = MAGSAC_K * MAGSAC_SIGMA_MAX, MAGSAC_CONF, MAGSAC_ITERS
    thr2 = float(threshold) ** 2
    if engine is not None:
        # batches of hypotheses on the device until the confidence criterion holds for the best inlier ratio so far
        best_mask, best_cnt, done, needed = np.zeros(n, bool), 0, 0, int(max_iters)
        while done < min(needed, int(max_iters)):
            mask = _ransac_on_device(engine, mkpts0, mkpts1, threshold, DEVICE_BATCH, seed + done)
            done += DEVICE_BATCH
            cnt = int(mask.sum())
            if cnt > best_cnt:
                best_cnt, best_mask = cnt, mask
            needed = _needed(confidence, best_cnt / n)
        return _finish(p0, p1, best_mask, thr2, method, n, enable_degeneracy_check, seed)
    rng = np.random.default_rng(seed)
    best_mask, best_cnt, it, needed = np.zeros(n, bool), 0, 0, max_iters
    while it < min(needed, max_iters):
        idx = rng.choice(n, 8, replace=False)
        try:
            F = _eight_point(p0[idx], p1[idx])
        except np.linalg.LinAlgError:
            it += 1
            continue
        mask = _sampson(F, p0, p1) < thr2
        cnt = int(mask.sum())
        if cnt > best_cnt:
            best_cnt, best_mask = cnt, mask
            needed = _needed(confidence, cnt / n)
        it += 1
    return _finish(p0, p1, best_mask, thr2, method, n, enable_degeneracy_check, seed)


def _weighted_eight_point(p0: np.ndarray, p1: np.ndarray, w: np.ndarray) -> np.ndarray:
    x0, T0 = _normalise(p0)
    x1, T1 = _normalise(p1)
    A = np.einsum("ni,nj->nij", x1, x0).reshape(len(p0), 9) * np.sqrt(w)[:, None]
    _, _, vt = np.linalg.svd(A, full_matrices=False)
    u, s, vt2 = np.linalg.svd(vt[-1].reshape(3, 3))
    F = T1.T @ (u @ np.diag([s[0], s[1], 0.0]) @ vt2) @ T0
    return F / max(np.linalg.norm(F), 1e-12)


def _magsac_weights(r: np.ndarray, sigma_max: float = MAGSAC_SIGMA_MAX, k: float = MAGSAC_K) -> np.ndarray:
    """MAGSAC++ weight of a residual r (Barath et al. 2020, eq. 9 with n = 4 degrees of freedom): the likelihood of the point
    being an inlier marginalised over the noise scale sigma in (0, sigma_max]; zero beyond k sigma_max."""
    from scipy.special import gammaincc, gamma
    g = gamma(1.5)
    w = g * (gammaincc(1.5, r ** 2 / (2.0 * sigma_max ** 2)) - gammaincc(1.5, k ** 2 / 2.0))
    return np.where(r <= k * sigma_max, np.maximum(w, 0.0), 0.0) / sigma_max


def _homography_dlt(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    xa, Ta = _normalise(a)
    xb, Tb = _normalise(b)
    rows = []
    for (x, y, _), (u, v, _) in zip(xa, xb):
        rows.append([0, 0, 0, -x, -y, -1, v * x, v * y, v])
        rows.append([x, y, 1, 0, 0, 0, -u * x, -u * y, -u])
    _, _, vt = np.linalg.svd(np.asarray(rows))
    return np.linalg.inv(Tb) @ vt[-1].reshape(3, 3) @ Ta


def _transfer_error(H: np.ndarray, p0: np.ndarray, p1: np.ndarray) -> np.ndarray:
    q = np.c_[p0, np.ones(len(p0))] @ H.T
    q = q[:, :2] / np.where(np.abs(q[:, 2:]) < 1e-12, 1e-12, q[:, 2:])
    return ((q - p1) ** 2).sum(1)


def _degeneracy_check(p0, p1, F, mask, thr2, rng):
    """DEGENSAC's repair of a plane-dominated solution: when one homography explains most inliers of F, estimate
    F = [e']x H by plane-and-parallax (H from the planar inliers, e' from two matches off the plane) and keep it if it
    explains more matches than F does."""
    inl = np.where(mask)[0]
    if len(inl) < 12:
        return F, mask
    bestH, bestm = None, None
    for _ in range(64):
        idx = rng.choice(inl, 4, replace=False)
        try:
            H = _homography_dlt(p0[idx], p1[idx])
        except np.linalg.LinAlgError:
            continue
        hm = _transfer_error(H, p0, p1) < 4.0 * thr2
        if bestm is None or hm.sum() > bestm.sum():
            bestH, bestm = H, hm
    if bestH is None or bestm[inl].sum() < 0.6 * len(inl):
        return F, mask                                                     # no dominant plane
    try:
        bestH = _homography_dlt(p0[bestm], p1[bestm])
    except np.linalg.LinAlgError:
        return F, mask
    off = np.where(~(_transfer_error(bestH, p0, p1) < 4.0 * thr2))[0]
    if len(off) < 2:
        return F, mask
    x1 = np.c_[p1, np.ones(len(p1))]
    hx0 = np.c_[p0, np.ones(len(p0))] @ bestH.T
    lines = np.cross(x1[off], hx0[off])                                    # each passes through the epipole e'
    best = (int(mask.sum()), F, mask)
    for _ in range(200):
        i, j = rng.choice(len(off), 2, replace=False)
        e = np.cross(lines[i], lines[j])
        if np.linalg.norm(e) < 1e-12:
            continue
        ex = np.array([[0, -e[2], e[1]], [e[2], 0, -e[0]], [-e[1], e[0], 0]])
        Fp = ex @ bestH
        Fp = Fp / max(np.linalg.norm(Fp), 1e-12)
        m = _sampson(Fp, p0, p1) < thr2
        if int(m.sum()) > best[0]:
            best = (int(m.sum()), Fp, m)
    if best[1] is not F:
        logger.info(f"Degeneracy check: plane-and-parallax model kept ({best[0]} inliers instead of {int(mask.sum())})")
    return best[1], best[2]


def _finish(p0, p1, best_mask, thr2, method, n, degeneracy_check=True, seed=0):
    """Local optimisation of the best hypothesis (and, per method, the degeneracy repair / the sigma-consensus refit)."""
    if int(best_mask.sum()) < 8:
        logger.error("Geometric verification failed: all matches kept")
        return None, np.ones(n, dtype=bool)
    F, mask = None, best_mask
    for _ in range(10):                                    # LO: fit on the inliers, re-score, until no more inliers join
        Fn = _eight_point(p0[mask], p1[mask])
        mn = _sampson(Fn, p0, p1) < thr2
        if F is not None and mn.sum() <= mask.sum():
            break
        if mn.sum() < 8:
            break
        F, mask = Fn, mn
    if F is None:
        F = _eight_point(p0[best_mask], p1[best_mask])
        mask = best_mask
    if method == GeometricVerification.MAGSAC:
        for _ in range(5):                                 # sigma-consensus: iteratively re-weighted least squares
            w = _magsac_weights(np.sqrt(_sampson(F, p0, p1)))
            if (w > 0).sum() < 8:
                break
            F = _weighted_eight_point(p0, p1, w)
        mask = np.sqrt(_sampson(F, p0, p1)) <= MAGSAC_K * MAGSAC_SIGMA_MAX
    elif degeneracy_check:
        F, mask = _degeneracy_check(p0, p1, F, mask, thr2, np.random.default_rng(seed + 1))
    logger.info(f"Geometric verification ({method.name}): {int(mask.sum())}/{n} inliers")
    return F, mask


def _ransac_on_device(engine, mkpts0: np.ndarray, mkpts1: np.ndarray, threshold: float, n_hyp: int, seed: int) -> np.ndarray:
    """All hypotheses in one launch pair (`im_ransac_fundamental`); returns the inlier mask of the best one."""
    import torch
    from .._lib import ptr, stream_ptr
    dev = engine.device
    d0 = torch.from_numpy(np.ascontiguousarray(mkpts0, dtype=np.float32)).to(dev)
    d1 = torch.from_numpy(np.ascontiguousarray(mkpts1, dtype=np.float32)).to(dev)
    n = d0.shape[0]
    dF = torch.empty(9, dtype=torch.float64, device=dev)
    dmask = torch.empty(n, dtype=torch.uint8, device=dev)
    dinfo = torch.empty(2, dtype=torch.int32, device=dev)
    engine.ctx.call("im_ransac_fundamental", ptr(d0), ptr(d1), n, int(n_hyp), float(threshold), int(seed) & 0xFFFFFFFF,
                    ptr(dF), ptr(dmask), ptr(dinfo), stream_ptr())
    return dmask.cpu().numpy().astype(bool)
