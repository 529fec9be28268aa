"""Row f-4 of the scope table: the first SfM consumers of the matches - relative orientation of the stereo pair and
linear triangulation (`src/icepy4d/sfm/geometry.py:31-76`, `sfm/two_view_geometry.py:38-110`,
`sfm/triangulation.py:153-186`). The reference uses OpenCV (`findEssentialMat`, `recoverPose`), which is absent here, so
parity of `estimate_pose` is unpinned (same interface, same conventions, checked on synthetic geometry); the linear
triangulation is pure numpy in the reference and is reproduced to rounding (tests compare against a restatement of its
formulation). Small host-side linear algebra on S <= 1e4 matched points: not a device workload; the RANSAC inside
`estimate_pose` reuses the device hypothesis scorer of `geometric_verification` when an engine is passed."""
from typing import Optional, Tuple

import numpy as np

from .matching.enums import GeometricVerification
from .matching.geometric_verification import geometric_verification


def triangulate_nviews(P, ip) -> np.ndarray:
    """One point seen in n views (`triangulation.py:166-186`): P list of 3x4 projection matrices, ip list of homogeneous
    image points [x, y, 1]. Returns the homogeneous point normalised to X[3] = 1."""
    if len(ip) != len(P):
        raise ValueError("Number of points and number of cameras not equal.")
    rows = []
    for x, p in zip(ip, P):
        x = np.asarray(x, dtype=np.float64)
        p = np.asarray(p, dtype=np.float64)
        rows.append(x[0] * p[2] - x[2] * p[0])      # the cross product x x (P X) = 0, two independent rows per view
        rows.append(x[1] * p[2] - x[2] * p[1])
    A = np.asarray(rows)
    X = np.linalg.svd(A)[2][-1]
    return X / X[3]


def triangulate_points_linear(P1, P2, x1, x2) -> np.ndarray:
    """Two-view triangulation of n points (`triangulation.py:153-163`); x1, x2 are [n, 3] homogeneous image points.
    Vectorised DLT: one batched 4x4 SVD instead of the reference's Python loop over an (6 x 6) system per point."""
    x1, x2 = np.asarray(x1, np.float64), np.asarray(x2, np.float64)
    if len(x1) != len(x2):
        raise ValueError("Number of points don't match.")
    P1, P2 = np.asarray(P1, np.float64), np.asarray(P2, np.float64)
    A = np.stack([x1[:, 0:1] * P1[2] - x1[:, 2:3] * P1[0], x1[:, 1:2] * P1[2] - x1[:, 2:3] * P1[1],
                  x2[:, 0:1] * P2[2] - x2[:, 2:3] * P2[0], x2[:, 1:2] * P2[2] - x2[:, 2:3] * P2[1]], axis=1)   # [n, 4, 4]
    X = np.linalg.svd(A)[2][:, -1, :]
    return X / X[:, 3:4]


def _essential_from_fundamental(F: np.ndarray) -> np.ndarray:
    u, _, vt = np.linalg.svd(F)
    return u @ np.diag([1.0, 1.0, 0.0]) @ vt


def _recover_pose(E: np.ndarray, x0: np.ndarray, x1: np.ndarray, mask: np.ndarray):
    """The (R, t) of the four decompositions of E that puts most inliers in front of both cameras (what
    cv2.recoverPose does); x0, x1 are normalised image coordinates [n, 2]."""
    u, _, vt = np.linalg.svd(E)
    if np.linalg.det(u) < 0:
        u = -u
    if np.linalg.det(vt) < 0:
        vt = -vt
    W = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    h0 = np.c_[x0[mask], np.ones(int(mask.sum()))]
    h1 = np.c_[x1[mask], np.ones(int(mask.sum()))]
    P0 = np.eye(3, 4)
    best = (-1, None, None, None)
    for R in (u @ W @ vt, u @ W.T @ vt):
        for t in (u[:, 2], -u[:, 2]):
            P1 = np.c_[R, t]
            X = triangulate_points_linear(P0, P1, h0, h1)[:, :3]
            front = (X[:, 2] > 0) & ((X @ R.T + t)[:, 2] > 0)
            if int(front.sum()) > best[0]:
                best = (int(front.sum()), R, t, front)
    return best


def estimate_pose(kpts0: np.ndarray, kpts1: np.ndarray, K0: np.ndarray, K1: np.ndarray, thresh: float, conf: float = 0.9999,
                  engine=None, seed: int = 0) -> Optional[Tuple[np.ndarray, np.ndarray, np.ndarray]]:
    """`estimate_pose` of the reference (`sfm/geometry.py:31-76`): (R [3,3], t [3], inliers [n] bool) with
    x_cam1 = R x_cam0 + t, t up to scale; None with fewer than 5 matches. The reference runs cv2.findEssentialMat (5-point
    RANSAC) + cv2.recoverPose; here the epipolar geometry of the NORMALISED coordinates is estimated by the 8-point
    RANSAC of `geometric_verification` (on the device when `engine` is given), projected onto the essential manifold and
    decomposed with the cheirality test."""
    if len(kpts0) < 5:
        return None
    K0, K1 = np.asarray(K0, np.float64), np.asarray(K1, np.float64)
    f_mean = np.mean([K0[0, 0], K1[1, 1], K0[0, 0], K1[1, 1]])      # the reference's (sic) focal average, `geometry.py:60`
    norm_thresh = thresh / f_mean
    x0 = (np.asarray(kpts0, np.float64) - K0[[0, 1], [2, 2]][None]) / K0[[0, 1], [0, 1]][None]
    x1 = (np.asarray(kpts1, np.float64) - K1[[0, 1], [2, 2]][None]) / K1[[0, 1], [0, 1]][None]
    if len(x0) < 8:
        return None   # the 8-point solver needs 8 correspondences (the reference's 5-point solver would still run)
    F, mask = geometric_verification(x0.astype(np.float32), x1.astype(np.float32), GeometricVerification.PYDEGENSAC,
                                     threshold=norm_thresh, confidence=conf, seed=seed, engine=engine)
    if F is None:
        raise AssertionError("Unable to estimate Essential matrix")
    E = _essential_from_fundamental(F)
    n_front, R, t, front = _recover_pose(E, x0, x1, mask)
    inliers = mask.copy()
    inliers[np.flatnonzero(mask)[~front]] = False
    return R, t, inliers
