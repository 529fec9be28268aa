"""Geometric verification of matches by a fundamental-matrix RANSAC (row f-2 of the scope table: the step right
after the hot path). The reference delegates to pydegensac, falling back to OpenCV USAC_MAGSAC and finally to
"all inliers" (`src/icepy4d/matching/geometric_verification.py:55-100`); both are un-vendored C++ libraries that are
absent here, so PYDEGENSAC and MAGSAC are served by one seeded algorithm (normalised 8-point minimal solver on random
samples, Sampson error, final least-squares refit) in two forms: `engine=...` evaluates all hypotheses at once on the
device (`im_ransac_fundamental`, csrc/geometry.hip: what `match()` uses), without it the numpy loop below runs (the
checker of the device form in the tests, and what a host-only caller gets). Inlier sets are randomised algorithms'
outputs: parity with the reference is statistical only and unpinned."""
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


def geometric_verification(mkpts0: np.ndarray = None, mkpts1: np.ndarray = None,
                           method: GeometricVerification = GeometricVerification.PYDEGENSAC, threshold: float = 1,
                           confidence: float = 0.9999, max_iters: int = 10000, seed: int = 0, engine=None, **_ignored):
    """Returns (F [3,3] or None, inlier mask [S] bool), like the reference (`geometric_verification.py:11-102`)."""
    assert isinstance(method, GeometricVerification), "Invalid method. It must be a GeometricVerification enum"
    n = 0 if mkpts0 is None else len(mkpts0)
    if method == GeometricVerification.NONE or n < 8:
        if n < 8 and method != GeometricVerification.NONE:
            logger.warning("Not enough matches for geometric verification: all matches kept")
        return None, np.ones(n, dtype=bool)
    p0, p1 = np.asarray(mkpts0, np.float64), np.asarray(mkpts1, np.float64)
    thr2 = float(threshold) ** 2
    if engine is not None:
        best_mask = _ransac_on_device(engine, mkpts0, mkpts1, threshold, min(int(max_iters), 8192), seed)
        return _refit(p0, p1, best_mask, thr2, method, n)
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
            w = min(max(cnt / n, 1e-9), 1 - 1e-9)
            needed = int(np.ceil(np.log(1 - confidence) / np.log(1 - w ** 8)))
        it += 1
    return _refit(p0, p1, best_mask, thr2, method, n)


def _refit(p0: np.ndarray, p1: np.ndarray, best_mask: np.ndarray, thr2: float, method, n: int):
    """Least-squares 8-point on the inliers of the best hypothesis, final mask from the refitted matrix."""
    if int(best_mask.sum()) < 8:
        logger.error("Geometric verification failed: all matches kept")
        return None, np.ones(n, dtype=bool)
    F = _eight_point(p0[best_mask], p1[best_mask])
    mask = _sampson(F, p0, p1) < thr2
    if mask.sum() < 8:
        mask = best_mask
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
