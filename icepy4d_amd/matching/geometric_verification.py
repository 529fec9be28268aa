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

Hypotheses are generated and scored on the device (`im_ransac_fundamental`, csrc/geometry.hip: one launch pair per batch of
1024), in batches until the confidence criterion is met; `engine=...` is required - there is no host fallback. The numpy RANSAC
loop that checks the device form lives with the oracle (`oracle/gv_cpu.py`, tests only) and enters through `hypothesis_fn`.
The refinement of the winning hypothesis (local optimisation, degeneracy repair, sigma-consensus refit) is small host-side
linear algebra on the matched points, as in the libraries the reference calls."""
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


NEEDED_CAP = 1 << 30   # "keep sampling until max_iters": what the criterion says when (almost) nothing is an inlier yet


def _needed(confidence: float, w: float) -> int:
    """Hypotheses needed so that, with inlier ratio w, one all-inlier 8-point sample has been drawn with probability
    `confidence`: log(1 - confidence) / log(1 - w^8). For w below ~1 % the denominator underflows (w^8 < 1e-16): log1p keeps
    it exact down to the smallest positive double, and a zero denominator (w = 0) answers NEEDED_CAP instead of dividing."""
    w = min(max(float(w), 0.0), 1 - 1e-9)
    den = np.log1p(-(w ** 8))
    if not den < 0.0:
        return NEEDED_CAP
    need = np.log(max(1 - confidence, 1e-12)) / den
    return int(min(np.ceil(need), NEEDED_CAP))


def geometric_verification(mkpts0: np.ndarray = None, mkpts1: np.ndarray = None,
                           method: GeometricVerification = GeometricVerification.PYDEGENSAC, threshold: float = 1,
                           confidence: float = 0.9999, max_iters: int = 10000, seed: int = 0, engine=None,
                           enable_degeneracy_check: bool = True, hypothesis_fn=None, **_ignored):
    """Returns (F [3,3] or None, inlier mask [S] bool), like the reference (`geometric_verification.py:11-102`).
    `hypothesis_fn(n_hypotheses, seed) -> inlier mask of the best of them` replaces the device stage (test seam: the oracle's
    numpy RANSAC); without it an engine is mandatory."""
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
    if hypothesis_fn is None:
        if engine is None:
            raise RuntimeError("geometric_verification needs engine=... : hypotheses are generated and scored on the device "
                               "(im_ransac_fundamental); there is no host fallback")
        hypothesis_fn = lambda n_hyp, sd: _ransac_on_device(engine, mkpts0, mkpts1, threshold, n_hyp, sd)   # noqa: E731
    # batches of hypotheses until the confidence criterion holds for the best inlier ratio so far
    best_mask, best_cnt, done, needed = np.zeros(n, bool), 0, 0, int(max_iters)
    while done < min(needed, int(max_iters)):
        mask = hypothesis_fn(DEVICE_BATCH, seed + done)
        done += DEVICE_BATCH
        cnt = int(mask.sum())
        if cnt > best_cnt:
            best_cnt, best_mask = cnt, mask
        needed = _needed(confidence, best_cnt / n)
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


def _dlt_rows(xa: np.ndarray, xb: np.ndarray) -> np.ndarray:
    """The 2 n x 9 DLT system of homogeneous, normalised correspondences xa -> xb (leading batch dimensions allowed)."""
    x, y = xa[..., 0], xa[..., 1]
    u, v = xb[..., 0], xb[..., 1]
    z, o = np.zeros_like(x), np.ones_like(x)
    r0 = np.stack([z, z, z, -x, -y, -o, v * x, v * y, v], -1)
    r1 = np.stack([x, y, o, z, z, z, -u * x, -u * y, -u], -1)
    return np.concatenate([r0, r1], -2)


def _homography_dlt(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    xa, Ta = _normalise(a)
    xb, Tb = _normalise(b)
    _, _, vt = np.linalg.svd(_dlt_rows(xa, xb), full_matrices=len(a) < 5)
    return np.linalg.inv(Tb) @ vt[-1].reshape(3, 3) @ Ta


def _apply3(M: np.ndarray, x: np.ndarray, y: np.ndarray, transpose: bool = False):
    """Rows of M x for homogeneous points (x, y, 1) and a stack M [B, 3, 3] (M^T x with transpose): three [B, n] planes by
    broadcasting (a K = 3 product is not worth a BLAS call)."""
    if transpose:
        M = np.swapaxes(M, -1, -2)
    return [M[:, i, 0, None] * x + M[:, i, 1, None] * y + M[:, i, 2, None] for i in range(3)]


def _transfer_error(H: np.ndarray, p0: np.ndarray, p1: np.ndarray) -> np.ndarray:
    """Squared transfer error of H (or of a stack of H: [B, 3, 3] -> [B, n])."""
    q0, q1, q2 = _apply3(H.reshape(-1, 3, 3), p0[:, 0], p0[:, 1])
    q2 = np.where(np.abs(q2) < 1e-12, 1e-12, q2)
    err = (q0 / q2 - p1[:, 0]) ** 2 + (q1 / q2 - p1[:, 1]) ** 2
    return err[0] if H.ndim == 2 else err


def _sampson_many(Fs: np.ndarray, p0: np.ndarray, p1: np.ndarray) -> np.ndarray:
    """Sampson errors of a stack of fundamental matrices: [B, 3, 3] -> [B, n]."""
    a0, a1, a2 = _apply3(Fs, p0[:, 0], p0[:, 1])                     # F x0
    b0, b1, _ = _apply3(Fs, p1[:, 0], p1[:, 1], transpose=True)      # F^T x1
    num = (p1[:, 0] * a0 + p1[:, 1] * a1 + a2) ** 2
    return num / np.maximum(a0 ** 2 + a1 ** 2 + b0 ** 2 + b1 ** 2, 1e-24)


def _degeneracy_check(p0, p1, F, mask, thr2, rng):
    """DEGENSAC's repair of a plane-dominated solution: when one homography explains most inliers of F, estimate
    F = [e']x H by plane-and-parallax (H from the planar inliers, e' from two matches off the plane) and keep it if it
    explains more matches than F does. The 64 plane hypotheses and the 200 epipole candidates are each scored as one stack."""
    inl = np.where(mask)[0]
    if len(inl) < 12:
        return F, mask
    Hs = []
    for _ in range(64):
        idx = rng.choice(inl, 4, replace=False)
        try:
            Hs.append(_homography_dlt(p0[idx], p1[idx]))
        except np.linalg.LinAlgError:
            continue
    if not Hs:
        return F, mask
    hms = _transfer_error(np.stack(Hs), p0, p1) < 4.0 * thr2             # [64, n]
    b = int(np.argmax(hms.sum(1)))                                       # first of the best, like a running maximum
    bestH, bestm = Hs[b], hms[b]
    if bestm[inl].sum() < 0.6 * len(inl):
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
    ij = np.array([rng.choice(len(off), 2, replace=False) for _ in range(200)])
    e = np.cross(lines[ij[:, 0]], lines[ij[:, 1]])                         # [200, 3]
    ok = np.linalg.norm(e, axis=1) >= 1e-12
    if not ok.any():
        return F, mask
    e = e[ok]
    ex = np.zeros((len(e), 3, 3))
    ex[:, 0, 1], ex[:, 0, 2], ex[:, 1, 0], ex[:, 1, 2], ex[:, 2, 0], ex[:, 2, 1] = -e[:, 2], e[:, 1], e[:, 2], -e[:, 0], -e[:, 1], e[:, 0]
    Fps = ex @ bestH
    Fps = Fps / np.maximum(np.linalg.norm(Fps, axis=(1, 2), keepdims=True), 1e-12)
    ms = _sampson_many(Fps, p0, p1) < thr2                                 # [<= 200, n]
    cnt = ms.sum(1)
    k = int(np.argmax(cnt))
    if int(cnt[k]) > int(mask.sum()):
        logger.info(f"Degeneracy check: plane-and-parallax model kept ({int(cnt[k])} inliers instead of {int(mask.sum())})")
        return Fps[k], ms[k]
    return F, mask


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
    from .._lib import ptr
    dev = engine.device
    d0 = torch.from_numpy(np.ascontiguousarray(mkpts0, dtype=np.float32)).to(dev)
    d1 = torch.from_numpy(np.ascontiguousarray(mkpts1, dtype=np.float32)).to(dev)
    n = d0.shape[0]
    dF = torch.empty(9, dtype=torch.float64, device=dev)
    dmask = torch.empty(n, dtype=torch.uint8, device=dev)
    dinfo = torch.empty(2, dtype=torch.int32, device=dev)
    engine.ctx.call("im_ransac_fundamental", ptr(d0), ptr(d1), n, int(n_hyp), float(threshold), int(seed) & 0xFFFFFFFF,
                    ptr(dF), ptr(dmask), ptr(dinfo), engine.stream_ptr())
    return dmask.cpu().numpy().astype(bool)
