"""Row f-4 of the scope table: the first SfM consumers of the matches - relative orientation of the stereo pair and
linear triangulation (`src/icepy4d/sfm/geometry.py:31-76`, `sfm/two_view_geometry.py:38-110`,
`sfm/triangulation.py:153-186`). The reference uses OpenCV (`findEssentialMat`, `recoverPose`), which is absent here, so
parity of `estimate_pose` is unpinned (same interface, same conventions, checked on synthetic geometry); the linear
triangulation is pure numpy in the reference: host and device path here solve the reference's own system and are compared with the
reference's OUTPUTS (tests/golden/g10_triangulation.npz, written by importing the reference module; oracle/sfm_cpu.py restates it). Small host-side linear algebra on S <= 1e4 matched points: not a device workload; the RANSAC inside
`estimate_pose(engine=...)` generates and scores essential-matrix hypotheses on the device (`im_ransac_essential`, csrc/geometry.hip)
and `triangulate_points_linear(engine=...)` triangulates on the device (`im_triangulate_linear`); the cheirality test and the 5-7
match case (five-point solver on every 5-subset) stay host numpy: one 3 x 3 matrix."""
from typing import Optional, Tuple

import numpy as np

from .matching.enums import GeometricVerification
from .matching.geometric_verification import geometric_verification


def _lift(P, ip) -> np.ndarray:
    """The reference's system for one point seen in n views (`triangulation.py:176-183`): unknowns X (4) and one depth per view,
    rows  P_i X - lambda_i x_i = 0  (3 n x (4 + n)); batched over leading dimensions of `ip`."""
    n = len(P)
    ip = [np.asarray(x, dtype=np.float64) for x in ip]
    lead = ip[0].shape[:-1]
    M = np.zeros(lead + (3 * n, 4 + n))
    for i in range(n):
        M[..., 3 * i:3 * i + 3, :4] = np.asarray(P[i], dtype=np.float64)
        M[..., 3 * i:3 * i + 3, 4 + i] = -ip[i]
    return M


def triangulate_nviews(P, ip) -> np.ndarray:
    """One point seen in n views (`triangulation.py:166-186`): P list of 3x4 projection matrices, ip list of homogeneous
    image points [x, y, 1]. Returns the homogeneous point normalised to X[3] = 1: the right singular vector of the smallest
    singular value of the reference's system (`_lift`), equal to the reference's result to the rounding of the SVD."""
    if len(ip) != len(P):
        raise ValueError("Number of points and number of cameras not equal.")
    X = np.linalg.svd(_lift(P, ip))[2][-1, :4]
    return X / X[3]


def triangulate_points_linear(P1, P2, x1, x2, engine=None) -> np.ndarray:
    """Two-view triangulation of n points (`triangulation.py:153-163`); x1, x2 are [n, 3] homogeneous image points. The reference's
    system per point (6 x 6: the point and one depth per view), solved for all points by ONE batched SVD instead of the reference's Python
    loop; with `engine=` on the device (`im_triangulate_linear`: one thread per point, one-sided Jacobi SVD in fp64). Both equal the reference's
    outputs (tests/golden/g10_triangulation.npz) to 1e-9 relative. Until round 5 both solved the four cross-product rows  x (P X) = 0
    instead - the same point on exact correspondences, a different least-squares problem on noisy ones (up to 5e-3 relative on 0.4 px noise)."""
    x1, x2 = np.asarray(x1, np.float64), np.asarray(x2, np.float64)
    if len(x1) != len(x2):
        raise ValueError("Number of points don't match.")
    P1, P2 = np.asarray(P1, np.float64), np.asarray(P2, np.float64)
    if engine is not None:
        import torch
        from ._lib import ptr
        n = len(x1)
        d1 = torch.from_numpy(np.ascontiguousarray(x1)).to(engine.device)
        d2 = torch.from_numpy(np.ascontiguousarray(x2)).to(engine.device)
        dX = torch.empty((n, 4), dtype=torch.float64, device=engine.device)
        p1, p2 = np.ascontiguousarray(P1.reshape(12)), np.ascontiguousarray(P2.reshape(12))
        engine.ctx.call("im_triangulate_linear", p1.ctypes.data, p2.ctypes.data, ptr(d1), ptr(d2), n, ptr(dX), engine.stream_ptr())
        return dX.cpu().numpy()
    if len(x1) == 0:
        return np.zeros((0, 4))
    X = np.linalg.svd(_lift([P1, P2], [x1, x2]))[2][:, -1, :4]
    return X / X[:, 3:4]


# ---- five-point relative pose (what cv2.findEssentialMat runs on a minimal sample; restated from the published algorithm:
# D. Nister, "An efficient solution to the five-point relative pose problem", PAMI 2004, in the action-matrix form of
# H. Stewenius, C. Engels, D. Nister, "Recent developments on direct relative orientation", ISPRS J. 2006)
_MONO = [(3, 0, 0), (2, 1, 0), (2, 0, 1), (1, 2, 0), (1, 1, 1), (1, 0, 2), (0, 3, 0), (0, 2, 1), (0, 1, 2), (0, 0, 3),
         (2, 0, 0), (1, 1, 0), (1, 0, 1), (0, 2, 0), (0, 1, 1), (0, 0, 2), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)]
_DEG = {d: [m for m in _MONO if sum(m) <= d] for d in (1, 2, 3)}


def _pmul(a: np.ndarray, da: int, b: np.ndarray, db: int) -> np.ndarray:
    """Product of two polynomials in (x, y, z) given as coefficient arrays [..., 4, 4, 4] (exponent of x, y, z) of total degree
    da, db (da + db <= 3); leading batch dimensions broadcast."""
    out = np.zeros(np.broadcast_shapes(a.shape, b.shape))
    for ma in _DEG[da]:
        ca = a[..., ma[0], ma[1], ma[2]]
        for mb in _DEG[db]:
            out[..., ma[0] + mb[0], ma[1] + mb[1], ma[2] + mb[2]] += ca * b[..., mb[0], mb[1], mb[2]]
    return out


def essential_five_point(x0: np.ndarray, x1: np.ndarray) -> np.ndarray:
    """All real essential matrices (up to 10) consistent with FIVE correspondences in normalised image coordinates
    (x1^T E x0 = 0): [m, 3, 3], each scaled to unit Frobenius norm; m = 0 for a degenerate sample."""
    x0, x1 = np.asarray(x0, np.float64), np.asarray(x1, np.float64)
    assert x0.shape == (5, 2) and x1.shape == (5, 2)
    h0, h1 = np.c_[x0, np.ones(5)], np.c_[x1, np.ones(5)]
    A = np.einsum("ni,nj->nij", h1, h0).reshape(5, 9)
    nullv = np.linalg.svd(A)[2][5:]                                   # E = x E1 + y E2 + z E3 + E4
    Eb = nullv.reshape(4, 3, 3)
    E = np.zeros((3, 3, 4, 4, 4))                                     # entries of E as linear polynomials
    for k, m in enumerate(((1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0))):
        E[:, :, m[0], m[1], m[2]] = Eb[k]
    # det(E) = 0
    def minor(r0, c0, r1, c1):
        return _pmul(E[r0, c0], 1, E[r1, c1], 1)
    det = (_pmul(E[0, 0], 1, minor(1, 1, 2, 2) - minor(1, 2, 2, 1), 2) - _pmul(E[0, 1], 1, minor(1, 0, 2, 2) - minor(1, 2, 2, 0), 2)
           + _pmul(E[0, 2], 1, minor(1, 0, 2, 1) - minor(1, 1, 2, 0), 2))
    # 2 E E^T E - trace(E E^T) E = 0
    EEt = np.zeros((3, 3, 4, 4, 4))
    for i in range(3):
        for j in range(3):
            EEt[i, j] = sum(_pmul(E[i, k], 1, E[j, k], 1) for k in range(3))
    tr = EEt[0, 0] + EEt[1, 1] + EEt[2, 2]
    eqs = [det]
    for i in range(3):
        for j in range(3):
            eqs.append(2.0 * sum(_pmul(EEt[i, k], 2, E[k, j], 1) for k in range(3)) - _pmul(tr, 2, E[i, j], 1))
    M = np.array([[e[m[0], m[1], m[2]] for m in _MONO] for e in eqs])         # 10 x 20
    try:
        B = np.linalg.solve(M[:, :10], M[:, 10:])                             # cubic monomials in terms of the lower ones
    except np.linalg.LinAlgError:
        return np.zeros((0, 3, 3))
    # action matrix of multiplication by x on the basis [x^2, xy, xz, y^2, yz, z^2, x, y, z, 1]
    At = np.zeros((10, 10))
    At[0:6] = -B[[0, 1, 2, 3, 4, 5]]              # x * {x^2, xy, xz, y^2, yz, z^2} = {x^3, x^2 y, x^2 z, x y^2, xyz, x z^2}
    At[6, 0] = At[7, 1] = At[8, 2] = At[9, 6] = 1.0    # x * {x, y, z, 1} = {x^2, xy, xz, x}
    w, V = np.linalg.eig(At)                       # x * b(solution) = At b(solution): the monomial vector is an eigenvector
    out = []
    for k in range(10):
        if abs(w[k].imag) > 1e-9 * max(1.0, abs(w[k])) or abs(V[9, k]) < 1e-14:
            continue
        v = (V[:, k] / V[9, k]).real
        Em = v[6] * Eb[0] + v[7] * Eb[1] + v[8] * Eb[2] + Eb[3]
        nrm = np.linalg.norm(Em)
        if np.isfinite(nrm) and nrm > 0:
            out.append(Em / nrm)
    return np.array(out).reshape(-1, 3, 3)


def _sampson_e(E: np.ndarray, x0: np.ndarray, x1: np.ndarray) -> np.ndarray:
    h0, h1 = np.c_[x0, np.ones(len(x0))], np.c_[x1, np.ones(len(x1))]
    Ex0, Etx1 = h0 @ E.T, h1 @ E
    return (h1 * Ex0).sum(1) ** 2 / np.maximum(Ex0[:, 0] ** 2 + Ex0[:, 1] ** 2 + Etx1[:, 0] ** 2 + Etx1[:, 1] ** 2, 1e-24)


def _estimate_pose_few(x0: np.ndarray, x1: np.ndarray, thr2: float):
    """5 <= n < 8 matches: every 5-subset through the five-point solver, the candidate with most inliers (then the smallest
    Sampson error sum, then the most points in front of both cameras) wins - what a RANSAC over minimal samples converges to."""
    from itertools import combinations
    best = None
    for idx in combinations(range(len(x0)), 5):
        for E in essential_five_point(x0[list(idx)], x1[list(idx)]):
            err = _sampson_e(E, x0, x1)
            mask = err < thr2
            n_front, R, t, front = _recover_pose(E, x0, x1, mask)
            key = (int(mask.sum()), n_front, -float(err[mask].sum()))
            if R is not None and (best is None or key > best[0]):
                inl = mask.copy()
                inl[np.flatnonzero(mask)[~front]] = False
                best = (key, R, t, inl)
    return None if best is None else best[1:]


def _essential_from_fundamental(F: np.ndarray) -> np.ndarray:
    u, _, vt = np.linalg.svd(F)
    return u @ np.diag([1.0, 1.0, 0.0]) @ vt


def _essential_ransac_on_device(engine, x0: np.ndarray, x1: np.ndarray, threshold: float, confidence: float, seed: int,
                                max_iters: int = 10000):
    """RANSAC over essential-matrix hypotheses generated and scored on the device (`im_ransac_essential`), in batches of 1024
    until the confidence criterion holds for the best inlier ratio; then one refit on the inliers of the winner (8-point on all
    of them, projected onto the essential manifold) that is kept if it does not lose inliers. Returns (E, inlier mask)."""
    import torch
    from ._lib import ptr
    from .matching.geometric_verification import DEVICE_BATCH, _eight_point, _needed, _sampson
    dev = engine.device
    d0 = torch.from_numpy(np.ascontiguousarray(x0, dtype=np.float32)).to(dev)
    d1 = torch.from_numpy(np.ascontiguousarray(x1, dtype=np.float32)).to(dev)
    n = len(x0)
    dE = torch.empty(9, dtype=torch.float64, device=dev)
    dmask = torch.empty(n, dtype=torch.uint8, device=dev)
    dinfo = torch.empty(2, dtype=torch.int32, device=dev)
    best = (0, None, None)
    done, needed = 0, int(max_iters)
    while done < min(needed, int(max_iters)):
        engine.ctx.call("im_ransac_essential", ptr(d0), ptr(d1), n, DEVICE_BATCH, float(threshold), (int(seed) + done) & 0xFFFFFFFF,
                        ptr(dE), ptr(dmask), ptr(dinfo), engine.stream_ptr())
        done += DEVICE_BATCH
        cnt = int(dinfo[0].item())
        if cnt > best[0]:
            best = (cnt, dE.cpu().numpy().reshape(3, 3).copy(), dmask.cpu().numpy().astype(bool))
        needed = _needed(confidence, best[0] / n)
    if best[1] is None:
        return None, np.zeros(n, bool)
    cnt, E, mask = best
    if cnt >= 8:
        E2 = _essential_from_fundamental(_eight_point(x0[mask], x1[mask]))
        m2 = _sampson(E2, x0, x1) < threshold ** 2
        if int(m2.sum()) >= cnt:
            E, mask = E2 / np.linalg.norm(E2), m2
    return E, mask


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
                  engine=None, seed: int = 0, hypothesis_fn=None) -> Optional[Tuple[np.ndarray, np.ndarray, np.ndarray]]:
    """`estimate_pose` of the reference (`sfm/geometry.py:31-76`): (R [3,3], t [3], inliers [n] bool) with
    x_cam1 = R x_cam0 + t, t up to scale; None with fewer than 5 matches. The reference runs cv2.findEssentialMat (5-point
    RANSAC) + cv2.recoverPose; here, with 8 or more matches, the epipolar geometry of the NORMALISED coordinates is estimated by
    an 8-point RANSAC whose hypotheses are projected onto the essential manifold and scored ON THE DEVICE (`engine=...`:
    `im_ransac_essential`; with a `hypothesis_fn` test seam instead: the F-matrix RANSAC of `geometric_verification`, projected
    afterwards) and decomposed with the cheirality test; with 5-7 matches the five-point solver (`essential_five_point`) runs on every
    5-subset."""
    if len(kpts0) < 5:
        return None
    K0, K1 = np.asarray(K0, np.float64), np.asarray(K1, np.float64)
    f_mean = np.mean([K0[0, 0], K1[1, 1], K0[0, 0], K1[1, 1]])      # the reference's (sic) focal average, `geometry.py:60`
    norm_thresh = thresh / f_mean
    x0 = (np.asarray(kpts0, np.float64) - K0[[0, 1], [2, 2]][None]) / K0[[0, 1], [0, 1]][None]
    x1 = (np.asarray(kpts1, np.float64) - K1[[0, 1], [2, 2]][None]) / K1[[0, 1], [0, 1]][None]
    if len(x0) < 8:
        # fewer matches than the 8-point hypotheses of the RANSAC below need: the five-point solver on every 5-subset
        return _estimate_pose_few(x0, x1, norm_thresh ** 2)
    if engine is not None and hypothesis_fn is None:
        # device path: essential-matrix hypotheses (8-point, projected onto the essential manifold) generated and scored on the GPU
        E, mask = _essential_ransac_on_device(engine, x0, x1, norm_thresh, conf, seed)
        if E is None:
            raise AssertionError("Unable to estimate Essential matrix")
    else:
        F, mask = geometric_verification(x0.astype(np.float32), x1.astype(np.float32), GeometricVerification.PYDEGENSAC,
                                         threshold=norm_thresh, confidence=conf, seed=seed, engine=engine, hypothesis_fn=hypothesis_fn)
        if F is None:
            raise AssertionError("Unable to estimate Essential matrix")
        E = _essential_from_fundamental(F)
    n_front, R, t, front = _recover_pose(E, x0, x1, mask)
    inliers = mask.copy()
    inliers[np.flatnonzero(mask)[~front]] = False
    return R, t, inliers
