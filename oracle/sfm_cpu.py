"""*** TEST INFRASTRUCTURE: the oracle for row f-4 (triangulation). Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it. ***

CPU restatement of the reference's linear triangulation (`src/icepy4d/sfm/triangulation.py:153-186`), pinned by
tests/golden/g10_triangulation.npz, which `tools/gen_golden.py triangulation` writes by importing the reference module itself
(tests/test_oracle_golden.py::test_triangulation_oracle_equals_the_reference_golden).

The reference solves, per point seen in n views, the homogeneous system  [P_i | -x_i e_i] [X; lambda] = 0  (3 n x (4 + n): the projection
P_i X equals lambda_i x_i) by SVD and takes the right singular vector of the smallest singular value (`triangulation.py:176-186`); the
two-view function loops it over the points (`:153-163`). `estimate_pose` (`sfm/geometry.py:31-76`) is cv2.findEssentialMat / recoverPose: cv2 is
un-vendored and absent, that half of f-4 stays unpinned (tests check recovered poses against known ones).
"""
import numpy as np


def triangulate_nviews(P, ip):
    """`triangulation.py:166-186`: P list of 3 x 4 projection matrices, ip list of homogeneous image points; returns X / X[3]."""
    if not len(ip) == len(P):
        raise ValueError("Number of points and number of cameras not equal.")
    n = len(P)
    M = np.zeros((3 * n, 4 + n))
    for i in range(n):
        M[3 * i:3 * i + 3, :4] = np.asarray(P[i], dtype=np.float64)
        M[3 * i:3 * i + 3, 4 + i] = -np.asarray(ip[i], dtype=np.float64)
    V = np.linalg.svd(M)[-1]
    X = V[-1, :4]
    return X / X[3]


def triangulate_points_linear(P1, P2, x1, x2):
    """`triangulation.py:153-163`: two views, x1 / x2 [n, 3] homogeneous; one `triangulate_nviews` per point."""
    if not len(x2) == len(x1):
        raise ValueError("Number of points don't match.")
    return np.array([triangulate_nviews([P1, P2], [a, b]) for a, b in zip(x1, x2)])
