// Geometric verification of matches: fundamental-matrix RANSAC on the device (row f-2 of the scope table, the step
// right after the hot path; the reference delegates to pydegensac / OpenCV USAC_MAGSAC on the CPU,
// `src/icepy4d/matching/geometric_verification.py:55-100`).
//
// All hypotheses are evaluated at once, one thread each, in fp64:
//   sample 8 distinct correspondences (counter-based hash of (seed, hypothesis, draw): reproducible, order independent)
//   -> Hartley normalisation of the sample -> 8 x 9 system, null vector by Gauss-Jordan with full pivoting
//   -> rank 2 by removing the smallest right singular direction (Jacobi eigen-decomposition of F^T F)
//   -> denormalise, Frobenius-normalise -> count the correspondences with Sampson error < threshold^2.
// A second launch picks the hypothesis with the most inliers (ties: lowest index) and writes its matrix and mask.
// The final least-squares refit on the inliers stays on the host (one 3 x 3 matrix and S <= 1e4 points), exactly as
// in the numpy twin `icepy4d_amd/matching/geometric_verification.py` that the tests compare against.
// Integer / fp64 latency-bound work: no MFMA, a few hundred microseconds for 4096 hypotheses x 10^4 matches.
#include "ctx.h"

#include <cstdint>

namespace im {

__device__ __forceinline__ uint32_t rng_hash(uint32_t seed, uint32_t hyp, uint32_t draw) {
    uint32_t x = seed * 0x9E3779B9u ^ (hyp + 0x7F4A7C15u) * 0x85EBCA6Bu ^ (draw + 1u) * 0xC2B2AE35u;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

// symmetric 3 x 3 eigen-decomposition by cyclic Jacobi; returns the eigenvector of the smallest eigenvalue
__device__ void smallest_eigvec3(double a[3][3], double v[3]) {
    double q[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int r = p + 1; r < 3; ++r) {
                if (fabs(a[p][r]) < 1e-300) continue;
                const double theta = (a[r][r] - a[p][p]) / (2.0 * a[p][r]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {            // A <- A J
                    const double akp = a[k][p], akr = a[k][r];
                    a[k][p] = c * akp - s * akr; a[k][r] = s * akp + c * akr;
                }
                for (int k = 0; k < 3; ++k) {            // A <- J^T A
                    const double apk = a[p][k], ark = a[r][k];
                    a[p][k] = c * apk - s * ark; a[r][k] = s * apk + c * ark;
                }
                for (int k = 0; k < 3; ++k) {
                    const double qkp = q[k][p], qkr = q[k][r];
                    q[k][p] = c * qkp - s * qkr; q[k][r] = s * qkp + c * qkr;
                }
            }
    }
    int m = 0;
    if (a[1][1] < a[m][m]) m = 1;
    if (a[2][2] < a[m][m]) m = 2;
    for (int k = 0; k < 3; ++k) v[k] = q[k][m];
}

// symmetric N x N eigen-decomposition by cyclic Jacobi: on return a is (numerically) diagonal, the columns of q are eigenvectors
template <int N>
__device__ void jacobi_sym(double a[N][N], double q[N][N]) {
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) q[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 16; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < N - 1; ++p)
            for (int r = p + 1; r < N; ++r) off += fabs(a[p][r]);
        if (off < 1e-300) break;
        for (int p = 0; p < N - 1; ++p)
            for (int r = p + 1; r < N; ++r) {
                if (fabs(a[p][r]) < 1e-300) continue;
                const double theta = (a[r][r] - a[p][p]) / (2.0 * a[p][r]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < N; ++k) { const double akp = a[k][p], akr = a[k][r]; a[k][p] = c * akp - s * akr; a[k][r] = s * akp + c * akr; }
                for (int k = 0; k < N; ++k) { const double apk = a[p][k], ark = a[r][k]; a[p][k] = c * apk - s * ark; a[r][k] = s * apk + c * ark; }
                for (int k = 0; k < N; ++k) { const double qkp = q[k][p], qkr = q[k][r]; q[k][p] = c * qkp - s * qkr; q[k][r] = s * qkp + c * qkr; }
            }
    }
}

// Projection of a 3 x 3 matrix onto the essential manifold (two equal singular values, the third zero): with F = U diag(s) V^T,
// E = U diag(m, m, 0) V^T, m = (s1 + s2) / 2 - written as E = F V diag(m / s1, m / s2, 0) V^T from the eigen-decomposition of
// F^T F (what `cv2.findEssentialMat` guarantees of its result, `sfm/geometry.py:64-66`). Returns false for a rank < 2 input.
__device__ bool project_essential(double F[9]) {
    double M[3][3], V[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[i][j] = F[i] * F[j] + F[3 + i] * F[3 + j] + F[6 + i] * F[6 + j];
    jacobi_sym<3>(M, V);
    int o[3] = {0, 1, 2};                                  // eigenvalues in descending order
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (M[o[j]][o[j]] > M[o[i]][o[i]]) { const int t = o[i]; o[i] = o[j]; o[j] = t; }
    const double s1 = sqrt(fmax(M[o[0]][o[0]], 0.0)), s2 = sqrt(fmax(M[o[1]][o[1]], 0.0));
    if (!(s2 > 1e-12 * s1) || !(s1 > 0.0)) return false;
    const double m = 0.5 * (s1 + s2), d1 = m / s1, d2 = m / s2;
    double D[3][3];                                        // V diag(d1, d2, 0) V^T
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) D[i][j] = d1 * V[i][o[0]] * V[j][o[0]] + d2 * V[i][o[1]] * V[j][o[1]];
    double E[9], nrm = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            E[3 * i + j] = F[3 * i] * D[0][j] + F[3 * i + 1] * D[1][j] + F[3 * i + 2] * D[2][j];
            nrm += E[3 * i + j] * E[3 * i + j];
        }
    nrm = sqrt(nrm);
    if (!(nrm > 1e-300)) return false;
    for (int j = 0; j < 9; ++j) F[j] = E[j] / nrm;
    return true;
}

__device__ __forceinline__ bool sampson_inlier(const double F[9], double x0, double y0, double x1, double y1, double thr2) {
    const double fx0 = F[0] * x0 + F[1] * y0 + F[2], fx1 = F[3] * x0 + F[4] * y0 + F[5], fx2 = F[6] * x0 + F[7] * y0 + F[8];
    const double ft0 = F[0] * x1 + F[3] * y1 + F[6], ft1 = F[1] * x1 + F[4] * y1 + F[7];
    const double num = x1 * fx0 + y1 * fx1 + fx2;
    const double den = fx0 * fx0 + fx1 * fx1 + ft0 * ft0 + ft1 * ft1;
    return num * num < thr2 * fmax(den, 1e-24);
}

// normalised 8-point on the sample idx[0..7]; false if the sample is degenerate
__device__ bool eight_point(const float* __restrict__ p0, const float* __restrict__ p1, const int idx[8], double F[9]) {
    double x0[8], y0[8], x1[8], y1[8];
    double c0x = 0, c0y = 0, c1x = 0, c1y = 0;
    for (int i = 0; i < 8; ++i) {
        x0[i] = p0[2 * idx[i]]; y0[i] = p0[2 * idx[i] + 1]; x1[i] = p1[2 * idx[i]]; y1[i] = p1[2 * idx[i] + 1];
        c0x += x0[i]; c0y += y0[i]; c1x += x1[i]; c1y += y1[i];
    }
    c0x *= 0.125; c0y *= 0.125; c1x *= 0.125; c1y *= 0.125;
    double d0 = 0, d1 = 0;
    for (int i = 0; i < 8; ++i) {
        d0 += sqrt((x0[i] - c0x) * (x0[i] - c0x) + (y0[i] - c0y) * (y0[i] - c0y));
        d1 += sqrt((x1[i] - c1x) * (x1[i] - c1x) + (y1[i] - c1y) * (y1[i] - c1y));
    }
    const double s0 = 1.4142135623730951 / fmax(d0 * 0.125, 1e-12), s1 = 1.4142135623730951 / fmax(d1 * 0.125, 1e-12);
    double A[8][9];
    for (int i = 0; i < 8; ++i) {
        const double a0 = s0 * (x0[i] - c0x), b0 = s0 * (y0[i] - c0y), a1 = s1 * (x1[i] - c1x), b1 = s1 * (y1[i] - c1y);
        A[i][0] = a1 * a0; A[i][1] = a1 * b0; A[i][2] = a1; A[i][3] = b1 * a0; A[i][4] = b1 * b0; A[i][5] = b1;
        A[i][6] = a0; A[i][7] = b0; A[i][8] = 1.0;
    }
    // Gauss-Jordan with full pivoting; the column that is never chosen as a pivot is the free variable
    int colperm[9];
    for (int j = 0; j < 9; ++j) colperm[j] = j;
    for (int k = 0; k < 8; ++k) {
        int pr = k, pc = k;
        double best = 0.0;
        for (int i = k; i < 8; ++i)
            for (int j = k; j < 9; ++j)
                if (fabs(A[i][j]) > best) { best = fabs(A[i][j]); pr = i; pc = j; }
        if (best < 1e-12) return false;
        for (int j = 0; j < 9; ++j) { const double t = A[k][j]; A[k][j] = A[pr][j]; A[pr][j] = t; }
        for (int i = 0; i < 8; ++i) { const double t = A[i][k]; A[i][k] = A[i][pc]; A[i][pc] = t; }
        { const int t = colperm[k]; colperm[k] = colperm[pc]; colperm[pc] = t; }
        const double inv = 1.0 / A[k][k];
        for (int j = k; j < 9; ++j) A[k][j] *= inv;
        for (int i = 0; i < 8; ++i) {
            if (i == k) continue;
            const double f = A[i][k];
            if (f == 0.0) continue;
            for (int j = k; j < 9; ++j) A[i][j] -= f * A[k][j];
        }
    }
    double f9[9];
    f9[colperm[8]] = 1.0;
    for (int k = 0; k < 8; ++k) f9[colperm[k]] = -A[k][8];
    // rank 2: remove the smallest right singular direction, F <- F - (F v) v^T
    double G[3][3], M[3][3], v[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) G[i][j] = f9[3 * i + j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[i][j] = G[0][i] * G[0][j] + G[1][i] * G[1][j] + G[2][i] * G[2][j];
    smallest_eigvec3(M, v);
    for (int i = 0; i < 3; ++i) {
        const double fv = G[i][0] * v[0] + G[i][1] * v[1] + G[i][2] * v[2];
        for (int j = 0; j < 3; ++j) G[i][j] -= fv * v[j];
    }
    // denormalise: F = T1^T G T0 with T = [[s, 0, -s cx], [0, s, -s cy], [0, 0, 1]]
    double H[3][3];   // G T0
    for (int i = 0; i < 3; ++i) {
        H[i][0] = G[i][0] * s0; H[i][1] = G[i][1] * s0;
        H[i][2] = -G[i][0] * s0 * c0x - G[i][1] * s0 * c0y + G[i][2];
    }
    double nrm = 0.0;
    for (int j = 0; j < 3; ++j) {
        F[j] = s1 * H[0][j]; F[3 + j] = s1 * H[1][j];
        F[6 + j] = -s1 * c1x * H[0][j] - s1 * c1y * H[1][j] + H[2][j];
    }
    for (int j = 0; j < 9; ++j) nrm += F[j] * F[j];
    nrm = sqrt(nrm);
    if (!(nrm > 1e-300)) return false;
    for (int j = 0; j < 9; ++j) F[j] /= nrm;
    return true;
}

__global__ __launch_bounds__(64) void ransac_hypotheses_kernel(const float* __restrict__ p0, const float* __restrict__ p1, int n, int n_hyp,
                                                               uint32_t seed, double thr2, int* __restrict__ counts, double* __restrict__ Fs,
                                                               int essential) {
    const int h = blockIdx.x * 64 + threadIdx.x;
    if (h >= n_hyp) return;
    int idx[8];
    uint32_t draw = 0;
    for (int i = 0; i < 8; ++i) {
        bool fresh;
        do {
            idx[i] = (int)(rng_hash(seed, (uint32_t)h, draw++) % (uint32_t)n);
            fresh = true;
            for (int j = 0; j < i; ++j) fresh = fresh && idx[j] != idx[i];
        } while (!fresh);
    }
    double F[9];
    int cnt = 0;
    if (eight_point(p0, p1, idx, F) && (!essential || project_essential(F))) {
        for (int i = 0; i < n; ++i) cnt += sampson_inlier(F, p0[2 * i], p0[2 * i + 1], p1[2 * i], p1[2 * i + 1], thr2) ? 1 : 0;
    } else {
        for (int j = 0; j < 9; ++j) F[j] = 0.0;
    }
    counts[h] = cnt;
    for (int j = 0; j < 9; ++j) Fs[(long)h * 9 + j] = F[j];
}

// one block: arg-max of the inlier counts (ties: lowest hypothesis), then the inlier mask of the winner
__global__ __launch_bounds__(1024) void ransac_select_kernel(const float* __restrict__ p0, const float* __restrict__ p1, int n, int n_hyp,
                                                              double thr2, const int* __restrict__ counts, const double* __restrict__ Fs,
                                                              double* __restrict__ F_out, uint8_t* __restrict__ mask, int* __restrict__ info) {
    __shared__ unsigned long long best[1024];
    unsigned long long b = 0;
    for (int h = threadIdx.x; h < n_hyp; h += 1024) {
        const unsigned long long key = ((unsigned long long)(uint32_t)counts[h] << 32) | (uint32_t)(0x7FFFFFFF - h);
        b = key > b ? key : b;
    }
    best[threadIdx.x] = b;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s && best[threadIdx.x + s] > best[threadIdx.x]) best[threadIdx.x] = best[threadIdx.x + s];
        __syncthreads();
    }
    const int hbest = 0x7FFFFFFF - (int)(uint32_t)(best[0] & 0xFFFFFFFFu);
    const int cbest = (int)(best[0] >> 32);
    double F[9];
    for (int j = 0; j < 9; ++j) F[j] = Fs[(long)hbest * 9 + j];
    if (threadIdx.x == 0) {
        info[0] = cbest; info[1] = hbest;
        for (int j = 0; j < 9; ++j) F_out[j] = F[j];
    }
    for (int i = threadIdx.x; i < n; i += 1024)
        mask[i] = sampson_inlier(F, p0[2 * i], p0[2 * i + 1], p1[2 * i], p1[2 * i + 1], thr2) ? 1 : 0;
}

static int ransac_entry(im_ctx* ctx, const char* who, int essential, const float* d_p0, const float* d_p1, int n, int n_hyp, double threshold,
                        unsigned int seed, double* d_F, uint8_t* d_mask, int32_t* d_info, void* stream) {
    IM_CHECK_CTX(ctx);
    if (n < 8 || n_hyp < 1) return ctx->fail(-70, "%s: needs >= 8 correspondences and >= 1 hypothesis", who);
    hipStream_t s = (hipStream_t)stream;
    int* counts = nullptr;
    double* Fs = nullptr;
    IM_HIP(ctx, hipMallocAsync((void**)&counts, sizeof(int) * n_hyp, s));
    IM_HIP(ctx, hipMallocAsync((void**)&Fs, sizeof(double) * 9 * n_hyp, s));
    hipLaunchKernelGGL(ransac_hypotheses_kernel, dim3((n_hyp + 63) / 64), dim3(64), 0, s, d_p0, d_p1, n, n_hyp, seed,
                       threshold * threshold, counts, Fs, essential);
    hipLaunchKernelGGL(ransac_select_kernel, dim3(1), dim3(1024), 0, s, d_p0, d_p1, n, n_hyp, threshold * threshold, counts, Fs, d_F,
                       d_mask, d_info);
    hipError_t e = hipGetLastError();
    hipFreeAsync(counts, s);
    hipFreeAsync(Fs, s);
    IM_HIP(ctx, e);
    return 0;
}

// Right singular vector of the smallest singular value of an N x N matrix by ONE-SIDED Jacobi (Hestenes): the columns of A are rotated in
// pairs until they are orthogonal, V collects the rotations, the singular values are the column norms. Works on A itself, not on A^T A: the
// accuracy follows the condition number, not its square (the triangulation system mixes entries of order 1e4 and of order 1).
template <int N>
__device__ void smallest_right_singular_vector(double a[N][N], double out[N]) {
    double v[N][N];
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) v[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < N - 1; ++p)
            for (int q = p + 1; q < N; ++q) {
                double al = 0.0, be = 0.0, ga = 0.0;
                for (int k = 0; k < N; ++k) { al += a[k][p] * a[k][p]; be += a[k][q] * a[k][q]; ga += a[k][p] * a[k][q]; }
                if (fabs(ga) <= 1e-17 * sqrt(al * be) || ga == 0.0) continue;
                rotated = true;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int k = 0; k < N; ++k) { const double x = a[k][p], y = a[k][q]; a[k][p] = c * x - sn * y; a[k][q] = sn * x + c * y; }
                for (int k = 0; k < N; ++k) { const double x = v[k][p], y = v[k][q]; v[k][p] = c * x - sn * y; v[k][q] = sn * x + c * y; }
            }
        if (!rotated) break;
    }
    int m = 0;
    double best = -1.0;
    for (int j = 0; j < N; ++j) {
        double nn = 0.0;
        for (int k = 0; k < N; ++k) nn += a[k][j] * a[k][j];
        if (best < 0.0 || nn < best) { best = nn; m = j; }
    }
    for (int k = 0; k < N; ++k) out[k] = v[k][m];
}

// linear two-view triangulation, one thread per point, in the REFERENCE's formulation (`sfm/triangulation.py:153-186`): the unknowns are the
// point X and one depth per view, the system  [P_i | -x_i e_i] [X; lambda] = 0  (6 x 6 for two views), the solution the right singular vector
// of its smallest singular value, normalised to X[3] = 1. (Until round 5 this kernel solved the four cross-product rows x (P X) = 0 instead:
// the same point on exact correspondences, another least-squares problem on noisy ones - found when the outputs were first compared with the
// reference's own, tests/golden/g10_triangulation.npz.)
__global__ __launch_bounds__(64) void triangulate_linear_kernel(const double* __restrict__ P, const double* __restrict__ x0,
                                                                const double* __restrict__ x1, int n, double* __restrict__ X) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    double M[6][6];
    for (int v = 0; v < 2; ++v) {
        const double* p = P + 12 * v;
        const double* x = (v == 0 ? x0 : x1) + 3 * (long)i;
        for (int r = 0; r < 3; ++r) {
            for (int j = 0; j < 4; ++j) M[3 * v + r][j] = p[4 * r + j];
            M[3 * v + r][4 + v] = -x[r];
            M[3 * v + r][5 - v] = 0.0;
        }
    }
    double sol[6];
    smallest_right_singular_vector<6>(M, sol);
    for (int k = 0; k < 4; ++k) X[4 * (long)i + k] = sol[k] / sol[3];
}

}  // namespace im

using namespace im;

extern "C" int im_ransac_fundamental(im_ctx* ctx, const float* d_p0, const float* d_p1, int n, int n_hyp, double threshold,
                                     unsigned int seed, double* d_F, uint8_t* d_mask, int32_t* d_info, void* stream) {
    return ransac_entry(ctx, "im_ransac_fundamental", 0, d_p0, d_p1, n, n_hyp, threshold, seed, d_F, d_mask, d_info, stream);
}

extern "C" int im_ransac_essential(im_ctx* ctx, const float* d_x0, const float* d_x1, int n, int n_hyp, double threshold,
                                   unsigned int seed, double* d_E, uint8_t* d_mask, int32_t* d_info, void* stream) {
    return ransac_entry(ctx, "im_ransac_essential", 1, d_x0, d_x1, n, n_hyp, threshold, seed, d_E, d_mask, d_info, stream);
}

extern "C" int im_triangulate_linear(im_ctx* ctx, const double* h_P0, const double* h_P1, const double* d_x0, const double* d_x1, int n,
                                     double* d_X, void* stream) {
    IM_CHECK_CTX(ctx);
    if (!h_P0 || !h_P1 || !d_x0 || !d_x1 || !d_X || n < 0) return ctx->fail(-71, "im_triangulate_linear: null argument");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    double hP[24];
    for (int j = 0; j < 12; ++j) { hP[j] = h_P0[j]; hP[12 + j] = h_P1[j]; }
    double* dP = nullptr;
    IM_HIP(ctx, hipMallocAsync((void**)&dP, sizeof(hP), s));
    IM_HIP(ctx, hipMemcpyAsync(dP, hP, sizeof(hP), hipMemcpyHostToDevice, s));
    IM_HIP(ctx, hipStreamSynchronize(s));     // hP lives on this stack frame
    hipLaunchKernelGGL(triangulate_linear_kernel, dim3((n + 63) / 64), dim3(64), 0, s, dP, d_x0, d_x1, n, d_X);
    hipError_t e = hipGetLastError();
    hipFreeAsync(dP, s);
    IM_HIP(ctx, e);
    return 0;
}
