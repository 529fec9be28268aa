"""TEST INFRASTRUCTURE (oracle): numpy RANSAC over normalised 8-point hypotheses scored by Sampson error - the checker of the
device hypothesis stage `im_ransac_fundamental` (csrc/geometry.hip), which restates what the reference delegates to pydegensac /
cv2 (`src/icepy4d/matching/geometric_verification.py:55-100`; both libraries are absent here, so parity with their randomised
outputs is statistical only). Only tests/ may import this module; the product has no host fallback."""
import numpy as np


def _normalise(p):
    c = p.mean(0)
    d = np.sqrt(((p - c) ** 2).sum(1)).mean()
    s = np.sqrt(2.0) / max(d, 1e-12)
    T = np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])
    return np.c_[p, np.ones(len(p))] @ T.T, T


def eight_point(p0, p1):
    x0, T0 = _normalise(p0)
    x1, T1 = _normalise(p1)
    A = np.einsum("ni,nj->nij", x1, x0).reshape(len(p0), 9)
    F = np.linalg.svd(A, full_matrices=True)[2][-1].reshape(3, 3)
    u, s, vt = np.linalg.svd(F)
    F = T1.T @ (u @ np.diag([s[0], s[1], 0.0]) @ vt) @ T0
    return F / max(np.linalg.norm(F), 1e-12)


def sampson(F, p0, p1):
    x0, x1 = np.c_[p0, np.ones(len(p0))], np.c_[p1, np.ones(len(p1))]
    Fx0, Ftx1 = x0 @ F.T, x1 @ F
    return (x1 * Fx0).sum(1) ** 2 / np.maximum(Fx0[:, 0] ** 2 + Fx0[:, 1] ** 2 + Ftx1[:, 0] ** 2 + Ftx1[:, 1] ** 2, 1e-24)


def hypothesis_fn(mkpts0, mkpts1, threshold):
    """-> f(n_hypotheses, seed): inlier mask (Sampson error < threshold^2) of the best of n_hypotheses random 8-point models."""
    p0, p1 = np.asarray(mkpts0, np.float64), np.asarray(mkpts1, np.float64)
    thr2 = float(threshold) ** 2

    def run(n_hyp, seed):
        rng = np.random.default_rng(seed)
        best, best_cnt = np.zeros(len(p0), bool), -1
        for _ in range(int(n_hyp)):
            idx = rng.choice(len(p0), 8, replace=False)
            try:
                F = eight_point(p0[idx], p1[idx])
            except np.linalg.LinAlgError:
                continue
            mask = sampson(F, p0, p1) < thr2
            if int(mask.sum()) > best_cnt:
                best, best_cnt = mask, int(mask.sum())
            if best_cnt > 0.9 * len(p0):          # early exit: plenty for the caller's confidence criterion
                break
        return best
    return run
