// SuperGlue (`SuperGlue/models/superglue.py:250-305`): keypoint encoder, 18-layer attentional GNN, score GEMM,
// log-domain Sinkhorn, mutual filter. The GNN reuses the fp32-MFMA GEMM and flash-attention kernels: SuperGlue's
// head layout (channel c = d * 4 + head, `view(b, 64, 4, N)` `:111-114`) is absorbed into the packed weights
// (projection rows and merge columns are permuted to head-major once at load), BatchNorm (eval) is folded into
// the preceding 1x1 convolution. The Sinkhorn sweeps never materialise the (M+1) x (N+1) coupling matrix:
// the similarity matrix stays read-only in HBM and the dustbin row / column are the scalar `bin_score`.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "ctx.h"
#include "lg_misc.h"
#include "workspace.h"

using namespace im;

namespace im {

// keypoint encoder input (`superglue.py:64-71, 82-84`): [(x - W/2) / (0.7 max(W, H)), (y - H/2) / .., score, 0 x 29]
__global__ __launch_bounds__(256) void sg_kenc_input_kernel(const float* __restrict__ kpts, const float* __restrict__ scores,
                                                             int kmax, const LGState* __restrict__ st, float4 shapes,
                                                             float* __restrict__ inp) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int i = t >> 5, f = t & 31;
    if (i >= st->n[b]) return;
    const float hgt = b == 0 ? shapes.x : shapes.z, wid = b == 0 ? shapes.y : shapes.w;
    const float scaling = fmaxf(wid, hgt) * 0.7f;
    float v = 0.f;
    if (f == 0) v = (kpts[((long)b * kmax + i) * 2] - wid / 2.f) / scaling;
    else if (f == 1) v = (kpts[((long)b * kmax + i) * 2 + 1] - hgt / 2.f) / scaling;
    else if (f == 2) v = scores[(long)b * kmax + i];
    inp[((long)b * kmax + i) * 32 + f] = v;
}

// ---- log-domain Sinkhorn (`log_sinkhorn_iterations`, `superglue.py:152-160`), couplings Z = [[sim, a], [a, a]]
__device__ __forceinline__ float sg_norm(int m, int n) { return -logf((float)m + (float)n); }

// Both sweeps are single-pass, 16-byte-per-lane streaming reads of the score matrix with an online (running max,
// rescaled sum) log-sum-exp, so every iteration reads the 4 (M)(N) bytes exactly twice (once per sweep).
struct OnlineLSE {
    float m = -INFINITY, s = 0.f;
    __device__ __forceinline__ void add(float x) {
        const float mn = fmaxf(m, x);
        s = s * expf(m - mn) + expf(x - mn);
        m = mn;
    }
    __device__ __forceinline__ void add4(float a, float b, float c, float d) {
        const float mn = fmaxf(fmaxf(m, fmaxf(a, b)), fmaxf(c, d));
        s = s * expf(m - mn) + ((expf(a - mn) + expf(b - mn)) + (expf(c - mn) + expf(d - mn)));
        m = mn;
    }
    __device__ __forceinline__ void merge(float om, float os) {
        const float mn = fmaxf(m, om);
        if (mn == -INFINITY) return;
        s = s * expf(m - mn) + os * expf(om - mn);
        m = mn;
    }
};

// u[i] = log_mu[i] - logsumexp_j(Z[i][j] + v[j]), i in [0, m]; wave per row, float4 per lane
__global__ __launch_bounds__(256) void sinkhorn_row_kernel(const float* __restrict__ sim, int ld, const int* __restrict__ m_ptr,
                                                            const int* __restrict__ n_ptr, float alpha,
                                                            const float* __restrict__ v, float* __restrict__ u) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int m = *m_ptr, n = *n_ptr;
    if (i > m || m <= 0 || n <= 0) return;
    const float* p = sim + (long)i * ld;
    const bool bin_row = i == m;
    OnlineLSE acc;
    const int n4 = ((ld & 3) == 0) ? (n & ~3) : 0;  // vector part needs 16-byte aligned rows
    for (int j = lane * 4; j < n4; j += 256) {
        const float4 vv = *reinterpret_cast<const float4*>(v + j);
        float4 x = make_float4(alpha, alpha, alpha, alpha);
        if (!bin_row) x = *reinterpret_cast<const float4*>(p + j);
        acc.add4(x.x + vv.x, x.y + vv.y, x.z + vv.z, x.w + vv.w);
    }
    for (int j = n4 + lane; j < n; j += 64) acc.add((bin_row ? alpha : p[j]) + v[j]);
    if (lane == 0) acc.add(alpha + v[n]);  // dustbin column
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc.merge(__shfl_xor(acc.m, off), __shfl_xor(acc.s, off));
    if (lane == 0) {
        const float norm = sg_norm(m, n);
        const float log_mu = bin_row ? logf((float)n) + norm : norm;
        u[i] = log_mu - (logf(acc.s) + acc.m);
    }
}

static constexpr int SK_STRIP = 256;  // rows per block of the column sweep

// column sweep: block = 256 columns x SK_STRIP rows; thread = 4 adjacent columns (one float4), 4 row lanes per column
// group, so every wave instruction reads 1 KiB of one matrix row. Per-strip partial (max, sum) per column.
__global__ __launch_bounds__(256) void sinkhorn_col_partial_kernel(const float* __restrict__ sim, int ld, const int* __restrict__ m_ptr,
                                                                    const int* __restrict__ n_ptr, float alpha,
                                                                    const float* __restrict__ u, float2* __restrict__ part, int pstride) {
    __shared__ float2 red[4][256];
    const int m = *m_ptr, n = *n_ptr;
    const int i0 = blockIdx.y * SK_STRIP;
    const int jb = blockIdx.x * 256;
    if (jb > n || i0 >= m || n <= 0) return;
    const int i1 = min(i0 + SK_STRIP, m);
    const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int j = jb + cg * 4;
    OnlineLSE a0, a1, a2, a3;
    const bool vec = ((ld & 3) == 0) && (j + 3 < n);
    if (vec) {
        for (int i = i0 + rl; i < i1; i += 4) {
            const float4 x = *reinterpret_cast<const float4*>(sim + (long)i * ld + j);
            const float ui = u[i];
            a0.add(x.x + ui); a1.add(x.y + ui); a2.add(x.z + ui); a3.add(x.w + ui);
        }
    } else {
        for (int i = i0 + rl; i < i1; i += 4) {
            const float ui = u[i];
            const float* q = sim + (long)i * ld;
            if (j < n) a0.add(q[j] + ui); else if (j == n) a0.add(alpha + ui);
            if (j + 1 < n) a1.add(q[j + 1] + ui); else if (j + 1 == n) a1.add(alpha + ui);
            if (j + 2 < n) a2.add(q[j + 2] + ui); else if (j + 2 == n) a2.add(alpha + ui);
            if (j + 3 < n) a3.add(q[j + 3] + ui); else if (j + 3 == n) a3.add(alpha + ui);
        }
    }
    red[rl][cg * 4 + 0] = make_float2(a0.m, a0.s); red[rl][cg * 4 + 1] = make_float2(a1.m, a1.s);
    red[rl][cg * 4 + 2] = make_float2(a2.m, a2.s); red[rl][cg * 4 + 3] = make_float2(a3.m, a3.s);
    __syncthreads();
    const int col = jb + threadIdx.x;
    if (col <= n) {
        OnlineLSE t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t.merge(red[r][threadIdx.x].x, red[r][threadIdx.x].y);
        part[(long)blockIdx.y * pstride + col] = make_float2(t.m, t.s);
    }
}

__global__ __launch_bounds__(256) void sinkhorn_col_combine_kernel(const float2* __restrict__ part, int pstride,
                                                                    const int* __restrict__ m_ptr, const int* __restrict__ n_ptr,
                                                                    float alpha, const float* __restrict__ u, float* __restrict__ v,
                                                                    float* __restrict__ norm_out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int m = *m_ptr, n = *n_ptr;
    if (j > n || m <= 0 || n <= 0) return;
    const int ns = (m + SK_STRIP - 1) / SK_STRIP;
    OnlineLSE t;
    t.add(alpha + u[m]);  // dustbin row
    for (int s = 0; s < ns; ++s) {
        const float2 p = part[(long)s * pstride + j];
        t.merge(p.x, p.y);
    }
    const float norm = sg_norm(m, n);
    const float log_nu = (j == n) ? logf((float)m) + norm : norm;
    v[j] = log_nu - (logf(t.s) + t.m);
    if (j == 0) *norm_out = norm;
}

// ---------------------------------------------------------------------------------------------------------
// One Sinkhorn iteration with ONE read of the coupling matrix (the two-sweep form above reads it twice): a block walks its rows two at
// a time; a row lives in REGISTERS (512 threads x 32 columns = 16384), so after the row's log-sum-exp has produced u_i the same
// registers feed the column statistics. The kernels this form went through (round 2: one row per step with online column maxima, three
// exponentials per element, 281 us per iteration at 16385^2; round 4: two rows per step in the log2 domain, 222 us; 1024-thread variants)
// are kept as a record in tools/experiments/sinkhorn_retired_forms.hip.txt - they are not part of the library any more.
static constexpr int SKF_T = 512, SKF_Q = 8, SKF_MAXN = SKF_T * 4 * SKF_Q;   // 16384 columns
static constexpr float SKF_NEG = -3.0e38f;
static constexpr float SK_L2E = 1.4426950408889634f, SK_LN2 = 0.6931471805599453f;

// keeps the instruction scheduler from interleaving the unrolled column groups of a pass (it would hold the temporaries of all of
// them at once: 290 registers wanted, 80 spilled); a group's dozen instructions are enough to cover the LDS read they start with
#define IM_SK_FENCE() __builtin_amdgcn_sched_barrier(0)

// ---------------------------------------------------------------------------------------------------------
// ONE exponential per element (round 4).
// With u_i fresh from the row pass, exp(z_ij + u_i + v_j - norm) is the row-softmax value the row pass has just computed,
//     p_ij = exp((z_ij + v_j) - M_i) / S_i,          e^(z_ij + u_i + v_j) = mu_i p_ij,
// so the column update needs no exponential of its own:  logsumexp_i(z_ij + u_i) = -v_j + norm + log C_j,  C_j = sum_i w_i p_ij  (w = 1, n for the
// dustbin row), hence     v_j <- v_j - log C_j   (+ log m for the dustbin column).   All terms are <= 1: no running maximum, no rescale.
// The one thing the online-max form gives for free is lost: when every p_ij of a column underflows (all rows put less than 1e-38 there under the PREVIOUS v)
// C_j is 0. Such columns are listed by the combine kernel (C_j < 1e-30) and recomputed exactly, by a strided column read, by the LAST combine block to
// finish (an integer ticket, as in the attention kernel's split-KV merge) - two launches per iteration; round 4 had a third, a repair kernel that
// found an empty list on practically every launch.
template <int SK4_T, int SK4_Q>
__global__ __launch_bounds__(SK4_T, 1) void sinkhorn_fused4_kernel(const float* __restrict__ sim, int ld, const int* __restrict__ m_ptr,
                                                                   const int* __restrict__ n_ptr, float alpha, const float* __restrict__ v,
                                                                   float* __restrict__ u, float* __restrict__ csum, int pstride) {
    static_assert(SK4_T * SK4_Q * 4 == SKF_MAXN, "columns");
    constexpr int SK4_W = SK4_T / 64;
    extern __shared__ __attribute__((aligned(16))) float sk_lds[];     // v * log2(e) [SKF_MAXN], then red[2][waves] float4
    float4* sv = reinterpret_cast<float4*>(sk_lds);
    float4* red = reinterpret_cast<float4*>(sk_lds + SKF_MAXN);
    const int m = *m_ptr, n = *n_ptr;
    if (m <= 0 || n <= 0 || (int)blockIdx.x > m) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x;
    float cS[4 * SK4_Q];
    const unsigned voff = (unsigned)tid * 16u;
    {
        // v -> LDS (times log2 e, masked columns very negative): eight range-checked 16-byte loads in flight per lane. As `j < n ? v[j] : ..`
        // per element this was 32 dword loads behind 32 exec-masked branches, each waited for on its own: ~9 us of every launch (round 5)
        const __amdgpu_buffer_rsrc_t rv = gmake_rsrc(v, (unsigned)n * 4u);
        float4 vq[SK4_Q];
#pragma unroll
        for (int q = 0; q < SK4_Q; ++q) vq[q] = gbuf_load4(rv, voff, (unsigned)q * (SK4_T * 16u));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < SK4_Q; ++q) {
            const int j = q * (SK4_T * 4) + tid * 4;
            sv[q * SK4_T + tid] = make_float4(j < n ? vq[q].x * SK_L2E : SKF_NEG, j + 1 < n ? vq[q].y * SK_L2E : SKF_NEG,
                                              j + 2 < n ? vq[q].z * SK_L2E : SKF_NEG, j + 3 < n ? vq[q].w * SK_L2E : SKF_NEG);
#pragma unroll
            for (int e = 0; e < 4; ++e) cS[4 * q + e] = 0.f;
        }
    }
    const float bin2 = (alpha + v[n]) * SK_L2E;
    float bS = 0.f;                                   // dustbin column (thread 0)
    const float norm = sg_norm(m, n);
    float4 ra[SK4_Q], rb[SK4_Q], rc[SK4_Q], rd[SK4_Q];
    // a REAL row (< m) through a range-checked descriptor; a row past the last one gets a descriptor of zero records - the loads are issued all
    // the same and return zeros that nothing uses (weight 0 in `step`). No branch here: with the dustbin row filled in behind `if (row >= m)`
    // the two paths met in register copies right behind the loads, i.e. every prefetched row was waited for as soon as it had been requested.
    auto load_row = [&](int row, float4 (&buf)[SK4_Q]) {
        const __amdgpu_buffer_rsrc_t rs = gmake_rsrc(sim + (long)min(row, m - 1) * ld, row < m ? (unsigned)n * 4u : 0u);
#pragma unroll
        for (int q = 0; q < SK4_Q; ++q) buf[q] = gbuf_load4(rs, voff, (unsigned)q * (SK4_T * 16u));
    };
    // rows iA and iB = iA + G (hasB false: absent); on return A and B hold the exponentials exp2(t - lane maximum)
    auto step = [&](int iA, float4 (&A)[SK4_Q], float4 (&B)[SK4_Q], int parity, bool hasB) {
        const int iB = iA + G;
        float mA = SKF_NEG, mB = SKF_NEG;
#pragma unroll
        for (int q = 0; q < SK4_Q; ++q) {
            const float4 w = sv[q * SK4_T + tid];
            A[q] = make_float4(fmaf(A[q].x, SK_L2E, w.x), fmaf(A[q].y, SK_L2E, w.y), fmaf(A[q].z, SK_L2E, w.z), fmaf(A[q].w, SK_L2E, w.w));
            B[q] = make_float4(fmaf(B[q].x, SK_L2E, w.x), fmaf(B[q].y, SK_L2E, w.y), fmaf(B[q].z, SK_L2E, w.z), fmaf(B[q].w, SK_L2E, w.w));
            mA = fmaxf(fmaxf(mA, fmaxf(A[q].x, A[q].y)), fmaxf(A[q].z, A[q].w));
            mB = fmaxf(fmaxf(mB, fmaxf(B[q].x, B[q].y)), fmaxf(B[q].z, B[q].w));
            IM_SK_FENCE();
        }
        if (tid == 0) { mA = fmaxf(mA, bin2); mB = fmaxf(mB, bin2); }
        float sA = 0.f, sB = 0.f;
#pragma unroll
        for (int q = 0; q < SK4_Q; ++q) {
            A[q] = make_float4(__builtin_amdgcn_exp2f(A[q].x - mA), __builtin_amdgcn_exp2f(A[q].y - mA), __builtin_amdgcn_exp2f(A[q].z - mA),
                               __builtin_amdgcn_exp2f(A[q].w - mA));
            B[q] = make_float4(__builtin_amdgcn_exp2f(B[q].x - mB), __builtin_amdgcn_exp2f(B[q].y - mB), __builtin_amdgcn_exp2f(B[q].z - mB),
                               __builtin_amdgcn_exp2f(B[q].w - mB));
            sA += (A[q].x + A[q].y) + (A[q].z + A[q].w);
            sB += (B[q].x + B[q].y) + (B[q].z + B[q].w);
            IM_SK_FENCE();
        }
        float ebA = 0.f, ebB = 0.f;                   // dustbin column entries of the two rows (thread 0)
        if (tid == 0) { ebA = __builtin_amdgcn_exp2f(bin2 - mA); ebB = __builtin_amdgcn_exp2f(bin2 - mB); sA += ebA; sB += ebB; }
        const float wMA = wave_max(mA), wMB = wave_max(mB);
        const float wSA = wave_sum(sA * __builtin_amdgcn_exp2f(mA - wMA)), wSB = wave_sum(sB * __builtin_amdgcn_exp2f(mB - wMB));
        if (lane == 0) red[parity * SK4_W + wave] = make_float4(wMA, wSA, wMB, wSB);
        __syncthreads();
        const float4 r = red[parity * SK4_W + (lane & (SK4_W - 1))];
        float MA = r.x, MB = r.z;
#pragma unroll
        for (int o = SK4_W / 2; o > 0; o >>= 1) { MA = fmaxf(MA, __shfl_xor(MA, o)); MB = fmaxf(MB, __shfl_xor(MB, o)); }
        float SA = r.y * __builtin_amdgcn_exp2f(r.x - MA), SB = r.w * __builtin_amdgcn_exp2f(r.z - MB);
#pragma unroll
        for (int o = SK4_W / 2; o > 0; o >>= 1) { SA += __shfl_xor(SA, o); SB += __shfl_xor(SB, o); }
        if (tid == 0) {
            u[iA] = ((iA == m) ? logf((float)n) + norm : norm) - (__builtin_amdgcn_logf(SA) + MA) * SK_LN2;
            if (hasB) u[iB] = ((iB == m) ? logf((float)n) + norm : norm) - (__builtin_amdgcn_logf(SB) + MB) * SK_LN2;
        }
        // column sums of the row-normalised matrix: p = e * 2^(lane max - row max) / S, weight n for the dustbin row, 0 for an absent row
        const float fA = __builtin_amdgcn_exp2f(mA - MA) / SA * (iA == m ? (float)n : 1.f);
        const float fB = hasB ? __builtin_amdgcn_exp2f(mB - MB) / SB * (iB == m ? (float)n : 1.f) : 0.f;
#pragma unroll
        for (int q = 0; q < SK4_Q; ++q) {
            cS[4 * q + 0] = fmaf(B[q].x, fB, fmaf(A[q].x, fA, cS[4 * q + 0]));
            cS[4 * q + 1] = fmaf(B[q].y, fB, fmaf(A[q].y, fA, cS[4 * q + 1]));
            cS[4 * q + 2] = fmaf(B[q].z, fB, fmaf(A[q].z, fA, cS[4 * q + 2]));
            cS[4 * q + 3] = fmaf(B[q].w, fB, fmaf(A[q].w, fA, cS[4 * q + 3]));
        }
        if (tid == 0) bS = fmaf(ebB, fB, fmaf(ebA, fA, bS));
    };
    // the block's real rows i = blockIdx.x, + G, .. < m, two per step, the next step's two rows in flight behind the current step's arithmetic
    int i = blockIdx.x;
    int parity = 0;
    if (i < m) {
        load_row(i, ra);
        load_row(i + G, rb);
        for (;;) {
            // (the prefetch is unconditional - rows past the end come back as zeros through their empty descriptors: behind `if (more)` the
            // loaded and the not-loaded path met in 64 register copies right behind the loads, which waited for every one of them)
            const bool more = i + 2 * G < m;
            load_row(i + 2 * G, rc); load_row(i + 3 * G, rd);
            step(i, ra, rb, 0, i + G < m);
            parity = 1;
            i += 2 * G;
            if (!more) break;
            const bool more2 = i + 2 * G < m;
            load_row(i + 2 * G, ra); load_row(i + 3 * G, rb);
            step(i, rc, rd, 1, i + G < m);
            parity = 0;
            i += 2 * G;
            if (!more2) break;
        }
    }
    // the dustbin row (index m, every entry = alpha, weight n): the last row of the block that owns it, a step of its own without loads - the
    // same arithmetic in the same order as when it rode in the B slot of that block's last step (an absent B adds exactly 0 to every sum)
    if ((int)blockIdx.x == m % G) {
#pragma unroll
        for (int q = 0; q < SK4_Q; ++q) ra[q] = make_float4(alpha, alpha, alpha, alpha);
        step(m, ra, rb, parity, false);
    }
    // column partials of this block: whole 16-byte quads (pstride % 4 == 0), streamed past L2 (the combine kernel is the only reader, on
    // other CUs; 16.8 MB of dirty lines at the kernel boundary cost 3 us, MI355X_MICROARCH.md "boundary"). The dustbin column's sum sits
    // in the quad that holds column n (thread 0 broadcasts it through LDS), or behind the last quad when n is a multiple of 4.
    float* bS_slot = reinterpret_cast<float*>(red + 2 * SK4_W);
    __syncthreads();
    if (tid == 0) *bS_slot = bS;
    __syncthreads();
    const float bSv = *bS_slot;
    float* pp = csum + (long)blockIdx.x * pstride;
#pragma unroll
    for (int q = 0; q < SK4_Q; ++q) {
        const int j = q * (SK4_T * 4) + tid * 4;
        if (j <= n) {
            f32x4 o = {cS[4 * q], cS[4 * q + 1], cS[4 * q + 2], cS[4 * q + 3]};
            const int d = n - j;
            if (d == 0) o[0] = bSv; else if (d == 1) o[1] = bSv; else if (d == 2) o[2] = bSv; else if (d == 3) o[3] = bSv;
            __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(pp + j));
        }
    }
    if (tid == 0 && n >= SKF_MAXN) pp[n] = bSv;
}

// v_j <- v_j - log C_j (+ log m for the dustbin column) from the per-block column sums; columns whose sum underflowed go on the repair list,
// which the last block to arrive (integer ticket, self-resetting) works off: exact v_j = log_nu_j - logsumexp_i(z_ij + u_i) by a strided column
// read, one listed column at a time - nothing to do, normally. u is complete (the sweep kernel has finished), every v_j has one writer.
static constexpr int SK4_CB = 128;      // columns per combine block: a wave reads 512 contiguous bytes of one partial row per instruction
__global__ __launch_bounds__(256) void sinkhorn_fused4_combine_kernel(const float* __restrict__ csum, int pstride, int n_parts,
                                                                       const int* __restrict__ m_ptr, const int* __restrict__ n_ptr, float* __restrict__ v,
                                                                       float* __restrict__ norm_out, int* list, int* cnt, int* ticket, float c_min,
                                                                       const float* __restrict__ sim, int ld, float alpha, const float* __restrict__ u) {
    __shared__ float2 red[4][64];
    __shared__ float2 red2[4];
    __shared__ int s_last;
    const int m = *m_ptr, n = *n_ptr;
    if (m <= 0 || n <= 0 || (int)blockIdx.x * SK4_CB > n) return;     // n / SK4_CB + 1 blocks take part (and a ticket)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * SK4_CB + 2 * lane;                        // this lane's two columns (pstride is even: 8-byte aligned pairs)
    const int ns = min(n_parts, m + 1);
    // wave w sums the partial rows w, w + 4, ...: eight loads in flight per lane; the order of the additions is fixed (deterministic)
    float2 acc = make_float2(0.f, 0.f);
    if (j <= n) {
        const float* cp = csum + j;
        int s0 = wave;
        for (; s0 + 28 < ns; s0 += 32) {
            float2 t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = *reinterpret_cast<const float2*>(cp + (long)(s0 + 4 * k) * pstride);
#pragma unroll
            for (int k = 0; k < 8; ++k) { acc.x += t[k].x; acc.y += t[k].y; }
        }
        for (; s0 < ns; s0 += 4) {
            const float2 t = *reinterpret_cast<const float2*>(cp + (long)s0 * pstride);
            acc.x += t.x; acc.y += t.y;
        }
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && j <= n) {
        const float C[2] = {(red[0][lane].x + red[1][lane].x) + (red[2][lane].x + red[3][lane].x),
                            (red[0][lane].y + red[1][lane].y) + (red[2][lane].y + red[3][lane].y)};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int jc = j + e;
            if (jc > n) break;
            if (C[e] > c_min) v[jc] = v[jc] - logf(C[e]) + (jc == n ? logf((float)m) : 0.f);
            else {                                               // order does not matter: each entry is recomputed on its own
                list[atomicAdd(cnt, 1)] = jc;
                __threadfence();                                 // rare path only: the entry must be out of this XCD's L2 before the ticket is taken
            }                                                    // (a fence in EVERY block - an L2 write-back each - cost 29 us per launch at 16385 columns)
        }
        if (j == 0) *norm_out = sg_norm(m, n);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int old = atomicAdd(ticket, 1);
        s_last = old == n / SK4_CB;
        if (s_last) *ticket = 0;                                 // ready for the next launch
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    const int count = *reinterpret_cast<volatile int*>(cnt);
    for (int idx = 0; idx < count; ++idx) {
        const int jr = reinterpret_cast<volatile int*>(list)[idx];
        OnlineLSE t;
        for (int i = threadIdx.x; i <= m; i += 256) t.add(((i < m && jr < n) ? sim[(long)i * ld + jr] : alpha) + u[i]);
        const float wm = wave_max(t.m);
        const float wsum = wave_sum(wm == -INFINITY ? 0.f : t.s * expf(t.m - wm));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red2[threadIdx.x >> 6] = make_float2(wm, wsum);
        __syncthreads();
        if (threadIdx.x == 0) {
            OnlineLSE a;
            for (int w = 0; w < 4; ++w) a.merge(red2[w].x, red2[w].y);
            const float norm = sg_norm(m, n);
            v[jr] = ((jr == n) ? logf((float)m) + norm : norm) - (logf(a.s) + a.m);
        }
    }
    if (threadIdx.x == 0) *cnt = 0;                              // the list is empty again for the next iteration
}

// full (m+1) x (n+1) transport matrix, only for the stage entry point / tests
__global__ __launch_bounds__(256) void ot_materialize_kernel(const float* __restrict__ sim, int ld, int m, int n, float alpha,
                                                              const float* __restrict__ u, const float* __restrict__ v,
                                                              float* __restrict__ out) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)(m + 1) * (n + 1);
    if (t >= total) return;
    const int i = (int)(t / (n + 1)), j = (int)(t - (long)i * (n + 1));
    const float z = (i < m && j < n) ? sim[(long)i * ld + j] : alpha;
    out[t] = ((z + u[i]) + v[j]) - sg_norm(m, n);
}

}  // namespace im

static int sinkhorn(im_ctx* ctx, hipStream_t s, const float* sim, int ld, const int* m_ptr, const int* n_ptr, int m_max, int n_max,
                    float alpha, int iters, float* u, float* v, float* norm_out) {
    Workspace* ws = ctx->ws;
    IM_HIP(ctx, launch_zero_words(u, m_max + 1, s));
    IM_HIP(ctx, launch_zero_words(v, n_max + 1, s));
    const int nstrips = (m_max + SK_STRIP - 1) / SK_STRIP;
    const int pstride = (n_max + 4) & ~3;       // >= n_max + 1 floats, rows 16-byte aligned (the partials are written and read as vectors)
    // single-read form: needs the row in the registers of one block (n <= 16384), 16-byte aligned rows, and as many partial
    // strips as blocks (the workspace holds (K + 15) / 16 + 1 of them). Otherwise (and with IM_SINKHORN_TWO_SWEEP=1, the A/B switch):
    // round 1's row sweep + column sweep, two reads of the couplings per iteration, any size.
    static const bool two_sweep = getenv("IM_SINKHORN_TWO_SWEEP") && getenv("IM_SINKHORN_TWO_SWEEP")[0] == '1';
    const int max_parts = (ctx->max_kpts + 15) / 16;
    if (!two_sweep && n_max <= SKF_MAXN && (ld % 4) == 0 && (reinterpret_cast<uintptr_t>(sim) % 16) == 0 && max_parts >= 1 && iters > 0) {
        const int G4 = std::min(std::min(256, 2 * max_parts), m_max + 1);
        const size_t lds4 = (SKF_MAXN + 2 * 16 * 4) * sizeof(float);
        static size_t lo4[IM_MAX_DEVICES] = {0};
        IM_HIP(ctx, ensure_dyn_lds(reinterpret_cast<const void*>(&sinkhorn_fused4_kernel<512, 8>), lds4, lo4));
        float* csum = reinterpret_cast<float*>(ws->part);       // [G4][pstride] floats in the partials buffer (room for 2 x max_parts rows)
        int* list = ws->ridx;                                   // free until the assignment stage: [0 .. n] column list, then counter and ticket
        int* cnt = ws->ridx + (ctx->max_kpts + 1);
        IM_HIP(ctx, launch_zero_words(cnt, 2, s));
        // IM_SINKHORN_REPAIR_ALL=1 (tests): every column goes through the exact repair path
        static const float c_min = (getenv("IM_SINKHORN_REPAIR_ALL") && getenv("IM_SINKHORN_REPAIR_ALL")[0] == '1') ? 3.0e38f : 1e-30f;
        for (int it = 0; it < iters; ++it) {
            hipLaunchKernelGGL((sinkhorn_fused4_kernel<512, 8>), dim3(G4), dim3(512), lds4, s, sim, ld, m_ptr, n_ptr, alpha, v, u, csum, pstride);
            hipLaunchKernelGGL(sinkhorn_fused4_combine_kernel, dim3((n_max + SK4_CB) / SK4_CB), dim3(256), 0, s, csum, pstride, G4, m_ptr, n_ptr, v, norm_out,
                               list, cnt, cnt + 1, c_min, sim, ld, alpha, u);
        }
        IM_HIP(ctx, hipGetLastError());
        return 0;
    }
    for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(sinkhorn_row_kernel, dim3((m_max + 1 + 3) / 4), dim3(256), 0, s, sim, ld, m_ptr, n_ptr, alpha, v, u);
        hipLaunchKernelGGL(sinkhorn_col_partial_kernel, dim3((n_max + 1 + 255) / 256, nstrips), dim3(256), 0, s, sim, ld, m_ptr, n_ptr,
                           alpha, u, ws->part, pstride);
        hipLaunchKernelGGL(sinkhorn_col_combine_kernel, dim3((n_max + 1 + 255) / 256), dim3(256), 0, s, ws->part, pstride, m_ptr, n_ptr,
                           alpha, u, v, norm_out);
    }
    if (iters == 0)  // norm is still needed by the assignment
        hipLaunchKernelGGL(sinkhorn_col_combine_kernel, dim3((n_max + 1 + 255) / 256), dim3(256), 0, s, ws->part, pstride, m_ptr, n_ptr,
                           alpha, u, v, norm_out);
    IM_HIP(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------ weights
static const std::vector<float>* sg_find(im_ctx* ctx, const std::string& key, size_t numel) {
    auto it = ctx->host_w.find("superglue/" + key);
    if (it == ctx->host_w.end()) { ctx->fail(-20, "weights: missing tensor %s of model superglue", key.c_str()); return nullptr; }
    if (it->second.size() != numel) {
        ctx->fail(-21, "weights: tensor %s has %zu elements, expected %zu", key.c_str(), it->second.size(), numel);
        return nullptr;
    }
    return &it->second;
}

// fold eval-mode BatchNorm1d (eps 1e-5) into the preceding 1x1 conv: y = g (Wx + b - mean) / sqrt(var + eps) + beta
static int fold_bn(im_ctx* ctx, const std::string& bn, int c, int in, std::vector<float>& w, std::vector<float>& b) {
    const auto* g = sg_find(ctx, bn + ".weight", c);
    const auto* be = sg_find(ctx, bn + ".bias", c);
    const auto* mu = sg_find(ctx, bn + ".running_mean", c);
    const auto* var = sg_find(ctx, bn + ".running_var", c);
    if (!g || !be || !mu || !var) return -20;
    for (int o = 0; o < c; ++o) {
        const float sc = (*g)[o] / std::sqrt((*var)[o] + 1e-5f);
        for (int i = 0; i < in; ++i) w[(size_t)o * in + i] *= sc;
        b[o] = (b[o] - (*mu)[o]) * sc + (*be)[o];
    }
    return 0;
}

int finalize_superglue(im_ctx* ctx) {
    SuperGlueW& W = ctx->sg;
    static const int dims[6] = {3, 32, 64, 128, 256, 256};
    for (int l = 0; l < 5; ++l) {
        const int in = dims[l], out = dims[l + 1], inp = l == 0 ? 32 : in;
        const std::string key = "kenc.encoder." + std::to_string(3 * l);
        const auto* w = sg_find(ctx, key + ".weight", (size_t)out * in);
        const auto* b = sg_find(ctx, key + ".bias", out);
        if (!w || !b) return -20;
        std::vector<float> ww(*w), bb(*b);
        if (l < 4 && fold_bn(ctx, "kenc.encoder." + std::to_string(3 * l + 1), out, in, ww, bb)) return -20;
        std::vector<float> wp((size_t)out * inp, 0.f);
        for (int o = 0; o < out; ++o)
            for (int i = 0; i < in; ++i) wp[(size_t)o * inp + i] = ww[(size_t)o * in + i];
        W.kenc_w[l] = ctx->upload(wp);
        W.kenc_b[l] = ctx->upload(bb);
    }
    const int L = 18;
    std::vector<float> qkv_w((size_t)L * 768 * 256), qkv_b((size_t)L * 768), mg_w((size_t)L * 65536), mg_b((size_t)L * 256),
        m0_w((size_t)L * 512 * 512), m0_b((size_t)L * 512), m3_w((size_t)L * 256 * 512), m3_b((size_t)L * 256);
    for (int l = 0; l < L; ++l) {
        const std::string p = "gnn.layers." + std::to_string(l);
        for (int which = 0; which < 3; ++which) {
            const auto* w = sg_find(ctx, p + ".attn.proj." + std::to_string(which) + ".weight", 65536);
            const auto* b = sg_find(ctx, p + ".attn.proj." + std::to_string(which) + ".bias", 256);
            if (!w || !b) return -20;
            for (int h = 0; h < 4; ++h)
                for (int d = 0; d < 64; ++d) {
                    const int src = d * 4 + h, dst = which * 256 + h * 64 + d;
                    memcpy(&qkv_w[((size_t)l * 768 + dst) * 256], &(*w)[(size_t)src * 256], 256 * sizeof(float));
                    qkv_b[(size_t)l * 768 + dst] = (*b)[src];
                }
        }
        const auto* mw = sg_find(ctx, p + ".attn.merge.weight", 65536);
        const auto* mb = sg_find(ctx, p + ".attn.merge.bias", 256);
        if (!mw || !mb) return -20;
        for (int o = 0; o < 256; ++o)
            for (int h = 0; h < 4; ++h)
                for (int d = 0; d < 64; ++d) mg_w[((size_t)l * 256 + o) * 256 + h * 64 + d] = (*mw)[(size_t)o * 256 + d * 4 + h];
        memcpy(&mg_b[(size_t)l * 256], mb->data(), 256 * sizeof(float));
        const auto* w0 = sg_find(ctx, p + ".mlp.0.weight", 512 * 512);
        const auto* b0 = sg_find(ctx, p + ".mlp.0.bias", 512);
        const auto* w3 = sg_find(ctx, p + ".mlp.3.weight", 256 * 512);
        const auto* b3 = sg_find(ctx, p + ".mlp.3.bias", 256);
        if (!w0 || !b0 || !w3 || !b3) return -20;
        std::vector<float> ww(*w0), bb(*b0);
        if (fold_bn(ctx, p + ".mlp.1", 512, 512, ww, bb)) return -20;
        // `merge` feeds only mlp.0 (`superglue.py:104-116`: mlp(cat([x, message]))): fold it into the message half of the
        // (BatchNorm-folded) mlp.0 weights, mlp0([x | Wm a + bm]) = W0a x + (W0b Wm) a + (W0b bm + b0), accumulated in
        // double. One 256 -> 256 GEMM launch and the message round trip less per layer.
        {
            const float* Wm = &mg_w[(size_t)l * 65536];     // head-permuted merge weights [256 out][256 in (h * 64 + d)]
            const float* bm = &mg_b[(size_t)l * 256];
            std::vector<double> row(256);
            for (int n = 0; n < 512; ++n) {
                float* w0r = &ww[(size_t)n * 512 + 256];
                double bacc = bb[n];
                for (int k = 0; k < 256; ++k) row[k] = 0.0;
                for (int j = 0; j < 256; ++j) {
                    const double wj = w0r[j];
                    const float* wmr = Wm + (size_t)j * 256;
                    for (int k = 0; k < 256; ++k) row[k] += wj * (double)wmr[k];
                    bacc += wj * (double)bm[j];
                }
                for (int k = 0; k < 256; ++k) w0r[k] = (float)row[k];
                bb[n] = (float)bacc;
            }
        }
        memcpy(&m0_w[(size_t)l * 512 * 512], ww.data(), ww.size() * sizeof(float));
        memcpy(&m0_b[(size_t)l * 512], bb.data(), 512 * sizeof(float));
        memcpy(&m3_w[(size_t)l * 256 * 512], w3->data(), w3->size() * sizeof(float));
        memcpy(&m3_b[(size_t)l * 256], b3->data(), 256 * sizeof(float));
    }
    W.proj_w = ctx->upload(qkv_w); W.proj_b = ctx->upload(qkv_b);
    W.mlp0_w = ctx->upload(m0_w); W.mlp0_b = ctx->upload(m0_b);
    W.mlp3_w = ctx->upload(m3_w); W.mlp3_b = ctx->upload(m3_b);
    {
        std::vector<float> p0, p3;
        for (int l = 0; l < L; ++l) {
            const std::vector<float> a0 = pack_frag_weights(&m0_w[(size_t)l * 512 * 512], 512, 512), a3 = pack_frag_weights(&m3_w[(size_t)l * 256 * 512], 256, 512);
            p0.insert(p0.end(), a0.begin(), a0.end());
            p3.insert(p3.end(), a3.begin(), a3.end());
        }
        W.mlp0_wp = ctx->upload(p0); W.mlp3_wp = ctx->upload(p3);
        std::vector<float> pq;
        pq.reserve(qkv_w.size() * 3 / 2);
        for (int l = 0; l < L; ++l) {
            const std::vector<float> one = pack_frag_weights(&qkv_w[(size_t)l * 768 * 256], 768, 256);
            pq.insert(pq.end(), one.begin(), one.end());
        }
        W.proj_wp = ctx->upload(pq);
        if (!W.mlp0_wp || !W.mlp3_wp || !W.proj_wp) return ctx->fail(-22, "weights: upload failed");
    }
    const auto* fw = sg_find(ctx, "final_proj.weight", 65536);
    const auto* fb = sg_find(ctx, "final_proj.bias", 256);
    const auto* bs = sg_find(ctx, "bin_score", 1);
    if (!fw || !fb || !bs) return -20;
    W.fp_w = ctx->upload(*fw); W.fp_b = ctx->upload(*fb);
    W.bin_score = (*bs)[0];
    if (!W.proj_w || !W.mlp0_w || !W.mlp3_w || !W.fp_w) return ctx->fail(-22, "weights: upload failed");
    W.ready = true;
    return 0;
}

extern "C" {

int im_superglue_forward(im_ctx* ctx, const float* d_kpts, const float* d_scores, const float* d_desc, const int32_t* d_n,
                         const float* h_shape, const im_superglue_conf* conf, int32_t* d_matches, float* d_mscores,
                         int32_t* d_info, void* stream) {
    IM_CHECK_CTX(ctx);
    if (!ctx->sg.ready) return ctx->fail(-50, "im_superglue_forward: weights not finalized");
    Workspace* ws = ctx->ws;
    if (!ws) return ctx->fail(-51, "im_superglue_forward: call im_ctx_reserve first");
    hipStream_t s = (hipStream_t)stream;
    const SuperGlueW& W = ctx->sg;
    const int K = ctx->max_kpts, L = conf->n_layers;
    if (L < 0 || L > 18) return ctx->fail(-52, "im_superglue_forward: n_layers must be 0..18");
    const long xb = (long)K * 256;
    LGState* st = ws->st;
    IM_HIP(ctx, launch_lg_init(st, 2, d_n, ws->ind[0], ws->prune, K, K, d_matches, d_mscores, K, s));
    float* x = ws->x[0];
    GemmArgs base;
    base.m_max = K; base.m_ptr = st->n; base.batch = 2; base.bx = 1;   // one pair: pstride is irrelevant for z < 2
    {   // keypoint encoder; the last layer adds the visual descriptors: x = desc + kenc(kpts, scores) (`superglue.py:269-270`)
        float* inp = ws->h;  // [2][K][32]
        const float4 shapes = make_float4(h_shape[0], h_shape[1], h_shape[2], h_shape[3]);
        hipLaunchKernelGGL(sg_kenc_input_kernel, dim3((K * 32 + 255) / 256, 2), dim3(256), 0, s, d_kpts, d_scores, K, st, shapes, inp);
        static const int dims[6] = {32, 32, 64, 128, 256, 256};
        const float* src = inp;
        float* bufs[2] = {ws->msg, ws->att};
        for (int l = 0; l < 5; ++l) {
            GemmArgs g = base;
            g.A = src; g.a_bstride = (long)K * dims[l]; g.lda = dims[l];
            g.W = W.kenc_w[l]; g.ldw = dims[l]; g.bias = W.kenc_b[l]; g.N = dims[l + 1]; g.K = dims[l];
            if (l < 4) {
                g.C = bufs[l & 1]; g.c_bstride = (long)K * dims[l + 1]; g.ldc = dims[l + 1]; g.epi = EPI_BIAS_RELU;
            } else {
                g.C = x; g.c_bstride = xb; g.ldc = 256; g.R = d_desc; g.r_bstride = xb; g.ldr = 256; g.epi = EPI_BIAS_RESID;
            }
            IM_LAUNCH(ctx, "sg_kenc_gemm", s, launch_gemm(g, s));
            src = bufs[l & 1];
        }
    }
    for (int l = 0; l < L; ++l) {
        const bool cross = (l & 1) != 0;  // ['self', 'cross'] * 9 (`superglue.py:215`)
        {
            GemmArgs g = base;
            g.A = x; g.a_bstride = xb; g.lda = 256; g.W = W.proj_w + (long)l * 768 * 256; g.ldw = 256;
            g.bias = W.proj_b + (long)l * 768; g.N = 768; g.K = 256; g.epi = EPI_QKV_ROPE;  // no rotary tables => plain q/k/v
            g.q = ws->q; g.k = ws->k; g.v = ws->v; g.head_bstride = (long)K * 256; g.head_stride = (long)K * 64;
            const char* const tiled_env = getenv("IM_PROJ_TILED");      // A/B switch (read per call): 1 = the tiled GEMM of rounds 1-5 instead of the row-block kernel
            if (tiled_env && tiled_env[0] == '1') {
                IM_LAUNCH(ctx, "sg_qkv_gemm", s, launch_gemm(g, s));
            } else {
                g.wp = reinterpret_cast<const unsigned char*>(W.proj_wp) + (size_t)l * 768 * 256 * 6;
                IM_LAUNCH(ctx, "sg_qkv_gemm", s, launch_proj_rows(g, s));
            }
        }
        {
            AttnArgs at;
            at.q = ws->q; at.k = ws->k; at.v = ws->v; at.hstride = (long)K * 64; at.bstride = (long)K * 256;
            at.out = ws->att; at.out_bstride = xb; at.ldo = 256; at.n_ptr = st->n; at.n_max = K; at.batch = 2; at.heads = 4;
            at.cross = cross ? 1 : 0; at.scale = 0.125f;  // / dim ** .5, dim = 64 (`superglue.py:91`)
            at.part = ws->attn_part; at.counters = ws->attn_cnt; at.planes = ws->attn_planes; at.clock = ctx->clock_of(0);
            IM_LAUNCH(ctx, "attn_kv_planes", s, launch_attn_planes(at, s));
            IM_LAUNCH(ctx, cross ? "flash_attn_cross" : "flash_attn_self", s, launch_flash_attn(at, s));
        }
        static const bool unfused = getenv("IM_FFN_UNFUSED") && getenv("IM_FFN_UNFUSED")[0] == '1';   // A/B switch: two GEMM launches
        if (!unfused) {   // x += mlp.3(relu(bn(mlp.0([x | att])))) with merge and BatchNorm folded into mlp.0: one kernel (ffn_fused.hip)
            FfnArgs f;
            f.act = 1; f.x = x; f.x_bstride = xb; f.att = ws->att; f.att_bstride = xb;
            f.w0p = W.mlp0_wp + (long)l * 512 * 512 * 3 / 2; f.b0 = W.mlp0_b + (long)l * 512;
            f.w3p = W.mlp3_wp + (long)l * 256 * 512 * 3 / 2; f.b3 = W.mlp3_b + (long)l * 256;
            f.m_max = base.m_max; f.batch = base.batch; f.m_ptr = base.m_ptr; f.active = base.active; f.pstride = base.pstride;
            IM_LAUNCH(ctx, "sg_mlp_fused", s, launch_ffn_fused(f, s));
            continue;
        }
        {
            GemmArgs g = base;
            g.A = x; g.a_bstride = xb; g.lda = 256; g.A1 = ws->att; g.a1_bstride = xb; g.lda1 = 256; g.ksplit = 256;   // merge folded in
            g.W = W.mlp0_w + (long)l * 512 * 512; g.ldw = 512; g.bias = W.mlp0_b + (long)l * 512; g.N = 512; g.K = 512;
            g.C = ws->h; g.c_bstride = (long)K * 512; g.ldc = 512; g.epi = EPI_BIAS_RELU;
            IM_LAUNCH(ctx, "sg_mlp0_gemm", s, launch_gemm(g, s));
        }
        {
            GemmArgs g = base;
            g.A = ws->h; g.a_bstride = (long)K * 512; g.lda = 512; g.W = W.mlp3_w + (long)l * 256 * 512; g.ldw = 512;
            g.bias = W.mlp3_b + (long)l * 256; g.N = 256; g.K = 512;
            g.C = x; g.c_bstride = xb; g.ldc = 256; g.R = x; g.r_bstride = xb; g.ldr = 256; g.epi = EPI_BIAS_RESID;
            IM_LAUNCH(ctx, "sg_mlp3_gemm", s, launch_gemm(g, s));
        }
    }
    {
        GemmArgs g = base;
        g.A = x; g.a_bstride = xb; g.lda = 256; g.W = W.fp_w; g.ldw = 256; g.bias = W.fp_b; g.N = 256; g.K = 256;
        g.C = ws->md; g.c_bstride = xb; g.ldc = 256; g.epi = EPI_BIAS;
        IM_LAUNCH(ctx, "sg_proj_gemm", s, launch_gemm(g, s));
        GemmArgs sgm;
        sgm.m_max = K; sgm.m_ptr = &st->n[0]; sgm.n_ptr = &st->n[1]; sgm.batch = 1; sgm.pair_batched = 1; sgm.bx = 1;
        sgm.A = ws->md; sgm.lda = 256; sgm.W = ws->md + xb; sgm.ldw = 256; sgm.N = K; sgm.K = 256;
        sgm.C = ws->sim; sgm.ldc = K; sgm.alpha = 0.0625f;  // / 256 ** .5 (`superglue.py:280`)
        sgm.epi = EPI_BIAS; sgm.big_tile = 1;
        IM_LAUNCH(ctx, "score_gemm", s, launch_gemm(sgm, s));
    }
    const int voff = (K + 4) & ~3;  // keep v 16-byte aligned for the float4 sweeps
    float* u = ws->uv;
    float* v = ws->uv + voff;
    float* norm = ws->uv + 2 * voff;
    if (ctx->prof_on) {
        im_ctx::ProfEntry pe{"sinkhorn", ctx->prof_event(), ctx->prof_event()};
        hipEventRecord(pe.e0, s);
        int rc = sinkhorn(ctx, s, ws->sim, K, &st->n[0], &st->n[1], K, K, W.bin_score, conf->sinkhorn_iterations, u, v, norm);
        hipEventRecord(pe.e1, s);
        ctx->prof.push_back(pe);
        if (rc) return rc;
    } else {
        int rc = sinkhorn(ctx, s, ws->sim, K, &st->n[0], &st->n[1], K, K, W.bin_score, conf->sinkhorn_iterations, u, v, norm);
        if (rc) return rc;
    }
    AssignArgs a;
    a.mode = 1;
    a.n_pairs = 1;
    a.sim = ws->sim; a.ld = K; a.m_ptr = &st->n[0]; a.n_ptr = &st->n[1]; a.m_max = K; a.n_max = K;
    a.rmax = u; a.cmax = v; a.rlog = norm; a.clog = ws->clog; a.part = ws->part;
    a.ridx = ws->ridx; a.rval = ws->rval; a.cbest = ws->cbest; a.threshold = (float)conf->match_threshold;
    a.out_m0 = d_matches; a.out_m1 = d_matches + K; a.out_s0 = d_mscores; a.out_s1 = d_mscores + K;
    IM_LAUNCH(ctx, "assign", s, launch_assign(a, s));
    IM_HIP(ctx, launch_lg_select_layer(st, 1, 0, ws->sel, d_info, s));
    IM_GUARD_CHECK(ctx, s, "im_superglue_forward");
    return 0;
}

int im_log_optimal_transport(im_ctx* ctx, const float* d_scores, int m, int n, int ld, float bin_score, int iters, float* d_out,
                             void* stream) {
    IM_CHECK_CTX(ctx);
    Workspace* ws = ctx->ws;
    const int K = ctx->max_kpts;
    if (!ws || m > K || n > K || m < 1 || n < 1) return ctx->fail(-41, "im_log_optimal_transport: m, n must be 1..max_kpts");
    hipStream_t s = (hipStream_t)stream;
    const int mn[2] = {m, n};
    IM_HIP(ctx, hipMemcpyAsync(ws->st->n, mn, sizeof(mn), hipMemcpyHostToDevice, s));
    IM_HIP(ctx, hipStreamSynchronize(s));
    const int voff = (K + 4) & ~3;
    float* u = ws->uv;
    float* v = ws->uv + voff;
    float* norm = ws->uv + 2 * voff;
    int rc = sinkhorn(ctx, s, d_scores, ld, &ws->st->n[0], &ws->st->n[1], m, n, bin_score, iters, u, v, norm);
    if (rc) return rc;
    const long total = (long)(m + 1) * (n + 1);
    hipLaunchKernelGGL(ot_materialize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_scores, ld, m, n, bin_score, u, v, d_out);
    IM_HIP(ctx, hipGetLastError());
    IM_GUARD_CHECK(ctx, s, "im_log_optimal_transport");
    return 0;
}

}  // extern "C"
