// Flash-style fp32 attention on the f32-input matrix cores: the K x K (or M x N) attention matrix is never
// materialised (the reference materialises it on CPU: `lightglue/lightglue.py:120-123` via SDPA,
// `:200-210` explicit einsum/softmax, `SuperGlue/models/superglue.py:87-93`).
//
//   out[z][i][h*64 + d] = sum_j softmax_j(scale * q_z[h][i] . k_y[h][j]) * v_y[h][j][d],   y = cross ? z^1 : z
//
// With cross = 1 and q = k = the shared `to_qk` projection this is exactly LightGlue's bidirectional
// cross attention: softmax over the rows of sim for image 0, over its columns for image 1.
//
// Work split: one wave owns 32 query rows, a block is 4 waves = 128 queries of one (image, head); all four
// waves stream the same 64-key K/V tiles through LDS. Both products keep the QUERY on the MFMA lane:
//     S^T (keys x queries)  = K . Q^T     A = K tile from LDS (16-byte reads, permuted d order), B = Q in registers
//     O^T (d x queries)    += V^T . P^T   A = V tile from LDS, B = P = exp2(S^T - m) straight from the accumulator layout
// so the online-softmax statistics (running max m, running sum l) are one scalar per lane and rescaling the
// output accumulator is a per-lane multiply.
//
// Pipeline (one barrier per 64-key tile): K and V tiles live in two LDS rings of depth 2, K running one tile
// ahead of V. Iteration t issues the QK^T MFMAs of tile t+1 and, in their shadow, the softmax VALU work of tile t,
// then the PV MFMAs of tile t; global loads for K(t+3) / V(t+2) are in flight meanwhile (register staging).
// Loads beyond the live key count are clamped to the last valid row (their scores are masked to -inf in the
// last tile), so the main loop has no divergent branch.
#include "common.h"
#include "kernels.h"

namespace im {

static constexpr int KT = 64;       // keys per tile
static constexpr int KS = 68;       // K tile row stride (floats): conflict-free ds_read_b128
static constexpr int VS = 64;       // V tile row stride
static constexpr int ATTN_LDS_FLOATS = 2 * KT * KS + 2 * KT * VS;

// register staging of one 64 x 64 tile: 1024 float4, 4 per thread (thread -> row idx >> 4, 16-byte column idx & 15)
__device__ __forceinline__ float4 load_row4(const float* __restrict__ base, int row0, int nk, int idx) {
    const int row = min(row0 + (idx >> 4), nk - 1);
    return *reinterpret_cast<const float4*>(base + (long)row * 64 + (idx & 15) * 4);
}
#define IM_LOAD_TILE(r, base, row0)                       \
    r##0 = load_row4(base, row0, nk, tid);                \
    r##1 = load_row4(base, row0, nk, tid + 256);          \
    r##2 = load_row4(base, row0, nk, tid + 512);          \
    r##3 = load_row4(base, row0, nk, tid + 768);
#define IM_STORE_TILE(r, dst, stride)                                                                          \
    *reinterpret_cast<float4*>((dst) + (tid >> 4) * (stride) + (tid & 15) * 4) = r##0;                        \
    *reinterpret_cast<float4*>((dst) + ((tid >> 4) + 16) * (stride) + (tid & 15) * 4) = r##1;                 \
    *reinterpret_cast<float4*>((dst) + ((tid >> 4) + 32) * (stride) + (tid & 15) * 4) = r##2;                 \
    *reinterpret_cast<float4*>((dst) + ((tid >> 4) + 48) * (stride) + (tid & 15) * 4) = r##3;

// S^T for one 64-key tile: two 32-key halves, 32 MFMAs each; contraction order d = s (hh = 0) / 32 + s (hh = 1).
// The K fragments of step t+1 are requested from LDS before the MFMAs of step t are issued.
__device__ __forceinline__ void qk_tile(const float* __restrict__ sK, int c, int hh, const float (&qf)[32], f32x16& sa, f32x16& sb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { sa[r] = 0.f; sb[r] = 0.f; }
    const float* ka = sK + c * KS + hh * 32;
    const float* kb = ka + 32 * KS;
    float4 fa = *reinterpret_cast<const float4*>(ka);
    float4 fb = *reinterpret_cast<const float4*>(kb);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        float4 ga = fa, gb = fb;
        if (t < 7) {
            ga = *reinterpret_cast<const float4*>(ka + (t + 1) * 4);
            gb = *reinterpret_cast<const float4*>(kb + (t + 1) * 4);
        }
        sa = mfma32(fa.x, qf[4 * t + 0], sa); sb = mfma32(fb.x, qf[4 * t + 0], sb);
        sa = mfma32(fa.y, qf[4 * t + 1], sa); sb = mfma32(fb.y, qf[4 * t + 1], sb);
        sa = mfma32(fa.z, qf[4 * t + 2], sa); sb = mfma32(fb.z, qf[4 * t + 2], sb);
        sa = mfma32(fa.w, qf[4 * t + 3], sa); sb = mfma32(fb.w, qf[4 * t + 3], sb);
        fa = ga; fb = gb;
    }
}

// online softmax of one 64-key tile in the log2 domain; on return sa / sb hold P = exp2(s * c - m)
template <bool TAIL>
__device__ __forceinline__ void softmax_tile(f32x16& sa, f32x16& sb, int kb, int nk, int hh, float& m_run, float& l_run,
                                             f32x16& o0, f32x16& o1) {
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (TAIL) {
            if (kb + acc_row(r, hh) >= nk) sa[r] = -INFINITY;
            if (kb + 32 + acc_row(r, hh) >= nk) sb[r] = -INFINITY;
        }
        mx = fmaxf(mx, fmaxf(sa[r], sb[r]));
    }
#ifdef IM_ABL_NO_SOFTMAX
    asm volatile("" ::"v"(mx));
    return;
#endif
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float pa = __builtin_amdgcn_exp2f(sa[r] - m_new), pb = __builtin_amdgcn_exp2f(sb[r] - m_new);
        sa[r] = pa; sb[r] = pb;
        rs += pa + pb;
    }
    rs += __shfl_xor(rs, 32);
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
}

// O^T += V^T . P^T for one 64-key tile: register r of P is key acc_row(r, hh) of its half. V fragments are read
// from LDS in groups of four key rows, one group ahead of the MFMAs that consume them.
__device__ __forceinline__ void pv_tile(const float* __restrict__ sV, int c, int hh, const f32x16& pa, const f32x16& pb, f32x16& o0, f32x16& o1) {
    const float* vbase = sV + (4 * hh) * VS + c;   // acc_row(r, hh) = (r & 3) + 8 * (r >> 2) + 4 * hh
    float va[4], vb[4], na[4], nb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { va[q] = vbase[q * VS]; vb[q] = vbase[q * VS + 32]; }
#pragma unroll
    for (int g = 0; g < 8; ++g) {      // group g: half g >> 2, key rows 8 * (g & 3) + (0..3) (+ 4 hh)
        if (g < 7) {
            const float* vn = vbase + (32 * ((g + 1) >> 2) + 8 * ((g + 1) & 3)) * VS;
#pragma unroll
            for (int q = 0; q < 4; ++q) { na[q] = vn[q * VS]; nb[q] = vn[q * VS + 32]; }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 4 * (g & 3) + q;
            const float p = (g < 4) ? pa[r] : pb[r];
            o0 = mfma32(va[q], p, o0);
            o1 = mfma32(vb[q], p, o1);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { va[q] = na[q]; vb[q] = nb[q]; }
    }
}

__global__ __launch_bounds__(256, 2) void flash_attn_f32_kernel(AttnArgs a) {
    if (a.active && *a.active == 0) return;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    // XCD-aware block -> tile map: workgroups are dealt round-robin over the 8 XCDs (bid % 8), each with a private
    // 4 MB L2. The (image, head) index is the fastest-varying part of the linear block id, so with
    // heads * batch = 8 every query block of one (image, head) lands on the same XCD and that XCD's L2 holds exactly
    // one K/V set (2 MB at 4096 keys) instead of all eight.
    const int hz = a.heads * a.batch;
    const int bid = blockIdx.x;
    const int head = (bid % hz) % a.heads, z = (bid % hz) / a.heads;
    const int y = a.cross ? (z ^ 1) : z;
    const int nq = a.n_ptr ? a.n_ptr[z] : a.n_max;
    const int nk = a.n_ptr ? a.n_ptr[y] : a.n_max;
    const int qb = (bid / hz) * 128;
    if (qb >= nq || nk <= 0) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const int qrow = qb + wave * 32 + c;

    const float* Q = a.q + (long)z * a.bstride + (long)head * a.hstride;
    const float* K = a.k + (long)y * a.bstride + (long)head * a.hstride;
    const float* V = a.v + (long)y * a.bstride + (long)head * a.hstride;

    // Q fragment: lane (c, hh) keeps Q[qrow][32*hh + s], s = 0..31
    float qf[32];
    {
        const float* qp = Q + (long)min(qrow, nq - 1) * 64 + hh * 32;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float4 v4 = *reinterpret_cast<const float4*>(qp + t * 4);
            qf[4 * t + 0] = v4.x; qf[4 * t + 1] = v4.y; qf[4 * t + 2] = v4.z; qf[4 * t + 3] = v4.w;
        }
        // softmax in the log2 domain: exp(s x) = exp2(x * s * log2 e); the factor is folded into Q once
        const float c2 = a.scale * 1.4426950408889634f;
#pragma unroll
        for (int t = 0; t < 32; ++t) qf[t] *= c2;
    }

    f32x16 o0, o1, sa, sb, na, nb;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    const int nt = (nk + KT - 1) / KT;

    // ---- prologue: K0 -> LDS, S(0); K1, V0 -> LDS; K2, V1 in registers
    float4 rk0, rk1, rk2, rk3, rv0, rv1, rv2, rv3;
    float* const sK0 = smem;
    float* const sV0 = smem + 2 * KT * KS;
    IM_LOAD_TILE(rk, K, 0)
    IM_STORE_TILE(rk, sK0, KS)
    IM_LOAD_TILE(rk, K, KT)
    IM_LOAD_TILE(rv, V, 0)
    __syncthreads();
    qk_tile(sK0, c, hh, qf, sa, sb);
    IM_STORE_TILE(rk, sK0 + KT * KS, KS)
    IM_STORE_TILE(rv, sV0, VS)
    IM_LOAD_TILE(rk, K, 2 * KT)
    IM_LOAD_TILE(rv, V, KT)
    __syncthreads();

    // iteration t: stage K(t+2), V(t+1); prefetch K(t+3), V(t+2); QK^T of tile t+1 with the softmax of tile t in
    // its shadow (one basic block, so the scheduler can interleave MFMA and VALU); PV of tile t; one barrier
// Instruction-group pipeline of one step (LLVM sched_group_barrier; masks: 0x008 MFMA, 0x002 VALU, 0x100 DS read,
// 0x200 DS write, 0x020 VMEM read): staging stores + prefetch loads first, then the 64 QK^T MFMAs with the
// previous tile's softmax VALU work and the K-fragment reads in their shadow, then the 64 PV MFMAs with the
// V-fragment reads running two pairs ahead.
#ifndef IM_ATTN_NO_SCHED
#define IM_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#define IM_SCHED_QK_SOFTMAX                                                    \
    IM_SGB(0x200, 8); IM_SGB(0x020, 8);                                        \
    IM_SGB(0x100, 4);                                                          \
    _Pragma("unroll") for (int _i = 0; _i < 16; ++_i) {                        \
        IM_SGB(0x008, 4); IM_SGB(0x002, 14); IM_SGB(0x100, 1);                 \
    }                                                                          \
    IM_SGB(0x100, 4);                                                          \
    _Pragma("unroll") for (int _i = 0; _i < 32; ++_i) {                        \
        IM_SGB(0x008, 2); IM_SGB(0x100, 1);                                    \
    }
#else
#define IM_SCHED_QK_SOFTMAX
#endif
#ifdef IM_ABL_NO_STAGE
#define IM_ABL_STAGE(...) (void)kw; (void)vw;
#else
#define IM_ABL_STAGE(...) __VA_ARGS__
#endif
#ifdef IM_ABL_NO_BARRIER
#define IM_ABL_BARRIER
#else
#define IM_ABL_BARRIER __syncthreads();
#endif
#define IM_STEP(TAIL)                                                                   \
    {                                                                                   \
        float* const kw = sK0 + (t & 1) * (KT * KS);                                    \
        float* const kr = sK0 + ((t + 1) & 1) * (KT * KS);                              \
        float* const vw = sV0 + ((t + 1) & 1) * (KT * VS);                              \
        float* const vr = sV0 + (t & 1) * (KT * VS);                                    \
        IM_ABL_STAGE(IM_STORE_TILE(rk, kw, KS)                                          \
        IM_STORE_TILE(rv, vw, VS)                                                       \
        IM_LOAD_TILE(rk, K, (t + 3) * KT)                                               \
        IM_LOAD_TILE(rv, V, (t + 2) * KT))                                              \
        qk_tile(kr, c, hh, qf, na, nb);                                                 \
        softmax_tile<TAIL>(sa, sb, t * KT, nk, hh, m_run, l_run, o0, o1);           \
        pv_tile(vr, c, hh, sa, sb, o0, o1);                                             \
        IM_SCHED_QK_SOFTMAX                                                             \
        sa = na; sb = nb;                                                               \
        IM_ABL_BARRIER                                                                  \
    }
    int t = 0;
    for (; t < nt - 1; ++t) IM_STEP(false)
    IM_STEP(true)
#undef IM_STEP

    // ---- epilogue: lane (c, hh) holds query qrow, d = db*32 + 8*g + 4*hh + (0..3) in registers 4g..4g+3
    if (qrow < nq) {
        const float inv = 1.f / l_run;
        float* op = a.out + (long)z * a.out_bstride + (long)qrow * a.ldo + head * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 w0 = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            const float4 w1 = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
            *reinterpret_cast<float4*>(op + 8 * g + 4 * hh) = w0;
            *reinterpret_cast<float4*>(op + 32 + 8 * g + 4 * hh) = w1;
        }
    }
}

hipError_t launch_flash_attn(const AttnArgs& a, hipStream_t s) {
    if (a.n_max <= 0) return hipSuccess;
    const size_t lds = ATTN_LDS_FLOATS * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_attn_f32_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid(((a.n_max + 127) / 128) * a.heads * a.batch), block(256);
    hipLaunchKernelGGL(flash_attn_f32_kernel, grid, block, lds, s, a);
    return hipGetLastError();
}

}  // namespace im
