// Flash-style fp32 attention on the f32-input matrix cores: the K x K (or M x N) attention matrix is never
// materialised (the reference materialises it on CPU: `lightglue/lightglue.py:120-123` via SDPA,
// `:200-210` explicit einsum/softmax, `SuperGlue/models/superglue.py:87-93`).
//
//   out[z][i][h*64 + d] = sum_j softmax_j(scale * q_z[h][i] . k_y[h][j]) * v_y[h][j][d],   y = cross ? z^1 : z
//
// With cross = 1 and q = k = the shared `to_qk` projection this is exactly LightGlue's bidirectional
// cross attention: softmax over the rows of sim for image 0, over its columns for image 1.
//
// Work split: one wave owns 32 query rows; a block is 128 queries of one (image, head) x G key groups of 4 waves each
// (G = 2: 8 waves; group g takes every G-th 64-key tile and the partial (O, m, l) are merged through LDS at the end).
// The four waves of a group stream the same K/V tiles through LDS. Both products keep the QUERY on the MFMA lane:
//     S^T (keys x queries)  = K . Q^T     A = K tile from LDS (16-byte reads, permuted d order), B = Q in registers
//     O^T (d x queries)    += V^T . P^T   A = V tile from LDS, B = P = exp2(S^T - m) straight from the accumulator layout
// so the online-softmax statistics (reference max m, running sum l) are one scalar per lane and rescaling the
// output accumulator is a per-lane multiply.
//
// Pipeline (one barrier per step of G tiles): K and V tiles live in two LDS rings of depth 2, K running one step
// ahead of V. Step t issues the QK^T MFMAs of step t+1, the softmax of step t, then the PV MFMAs of step t; the loads
// for K(t+3) / V(t+2) are in flight meanwhile (register staging by buffer loads: rows past the live key count read as
// zero through the descriptor's range check and are masked to -inf in the last step, so the loop has no divergent
// branch and no address arithmetic). On gfx950 VALU / LDS instructions are NOT hidden behind fp32 MFMAs
// (tools/mfma_peak.hip), so everything next to the MFMAs is written for instruction count - see softmax_tile.
#include "common.h"
#include "kernels.h"
#include "sp_post.h"

#include <cstdlib>

namespace im {

static constexpr int KT = 64;       // keys per tile
static constexpr int KS = 68;       // K tile row stride (floats): conflict-free ds_read_b128
static constexpr int VS = 64;       // V tile row stride
static constexpr int ATTN_LDS_FLOATS = 2 * KT * KS + 2 * KT * VS;   // per key group

// register staging of one (64 G) x 64 tile by NT = 256 G threads: 4 float4 per thread (thread -> row idx >> 4,
// 16-byte column idx & 15). The loads are buffer loads: a wave-uniform descriptor sized to the live key rows, a
// loop-invariant 32-bit lane offset and a scalar tile offset - no per-load address arithmetic on the VALU (which
// would cost matrix-pipe time, see softmax_tile), and rows past the last key read as zero without any clamping.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, int rows) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    void* p = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, __builtin_amdgcn_readfirstlane(rows) * 256, 0x00020000);
}
#define IM_LOAD_TILE(r, rsrc, tile)                                                      \
    {                                                                                    \
        const unsigned so_ = (unsigned)(tile) * (G * KT * 256u);                         \
        r##0 = buf_load4(rsrc, voff, so_);                                               \
        r##1 = buf_load4(rsrc, voff, so_ + (NT / 16) * 256u);                            \
        r##2 = buf_load4(rsrc, voff, so_ + 2 * (NT / 16) * 256u);                        \
        r##3 = buf_load4(rsrc, voff, so_ + 3 * (NT / 16) * 256u);                        \
    }
#define IM_STORE_TILE(r, dst, stride)                                                                          \
    *reinterpret_cast<float4*>((dst) + (tid >> 4) * (stride) + (tid & 15) * 4) = r##0;                        \
    *reinterpret_cast<float4*>((dst) + ((tid >> 4) + NT / 16) * (stride) + (tid & 15) * 4) = r##1;            \
    *reinterpret_cast<float4*>((dst) + ((tid >> 4) + 2 * NT / 16) * (stride) + (tid & 15) * 4) = r##2;        \
    *reinterpret_cast<float4*>((dst) + ((tid >> 4) + 3 * NT / 16) * (stride) + (tid & 15) * 4) = r##3;

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

// online softmax of one 64-key tile in the log2 domain; on return sa / sb hold P = exp2(s * c - m).
//
// On gfx950 the fp32 MFMA and the ordinary VALU share issue cycles: `tools/mfma_peak.hip` shows every VALU
// instruction next to a v_mfma_f32_32x32x2_f32 stream costs its full 4 (exp2: 8) cycles of matrix-pipe time, with one
// or two waves per SIMD alike. The softmax is therefore written for instruction COUNT, not for overlap:
//   * max as nested maxima that the compiler fuses into v_max3_f32 (16 instructions for the 32 scores of a lane);
//   * subtraction and row sum on whole vectors, which lower to packed fp32 (v_pk_add_f32: two values per lane-op);
//   * the running max is a REFERENCE, only raised when some row of the wave exceeds it by more than 2^8: P then stays
//     <= 256 (harmless in fp32: l <= 4096 keys x 256) and the common step has no alpha, no rescale of O, no l * alpha.
//     Softmax is shift invariant, so O / l is the same quantity as with the exact running max.
static constexpr float M_SLACK = 8.f;

// Compiler-visible since round 3 (rounds 1-2 had `asm("v_max3_f32 ...")` / `asm("v_pk_add_f32 ...")` here): an asm VALU write next to
// MFMAs is invisible to the hazard recogniser - in conv_wino.hip that produced wrong results as soon as the register allocation
// changed. The compiler forms v_max3_f32 from the nested maxima and v_pk_add_f32 from the two-element operations;
// the packed adds stay packed only where no MFMA precedes them closely (see the scheduling fences in softmax_tile).
__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { return a + b; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { return a - b; }
#define IM_PAIR(v, i) f32x2{v[2 * (i)], v[2 * (i) + 1]}

template <bool TAIL>
__device__ __forceinline__ void softmax_tile(f32x16& sa, f32x16& sb, int kb, int nk, int hh, float& m_run, float& l_run,
                                             f32x16& o0, f32x16& o1) {
    if (TAIL) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (kb + acc_row(r, hh) >= nk) sa[r] = -INFINITY;
            if (kb + 32 + acc_row(r, hh) >= nk) sb[r] = -INFINITY;
        }
    }
    float mx0 = max3(sa[0], sb[0], sa[1]), mx1 = max3(sb[1], sa[2], sb[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) { mx0 = max3(mx0, sa[r], sb[r]); mx1 = max3(mx1, sa[r + 1], sb[r + 1]); }
    float mx = max3(mx0, mx1, fmaxf(sa[15], sb[15]));
#ifdef IM_ABL_NO_SOFTMAX
    asm volatile("" ::"v"(mx));
    return;
#endif
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (__builtin_amdgcn_ballot_w64(mx > m_run + M_SLACK) != 0) {     // wave-uniform: raise the reference, rescale
        asm volatile("" ::: "memory");                                 // keep this a branch: if-converted it costs 16 v_pk_mul per step
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);     // m_run = -inf at the first tile: alpha = 0
        l_run *= alpha;
        o0 *= alpha;
        o1 *= alpha;
        m_run = m_new;
    }
    // a key group whose every tile so far lies past the last key keeps m = -inf, l = 0 (only possible in the tail)
    const float m_use = (TAIL && m_run == -INFINITY) ? 0.f : m_run;
    const f32x2 mm = {m_use, m_use};
    f32x2 acc[4];
    // The exp2 / row-sum section is fenced off from the scheduler on both sides: interleaved with the MFMAs of the neighbouring
    // products, the compiler's late "unpack packed instructions in the shadow of an MFMA" peephole turns every v_pk_add_f32 that lands
    // behind an MFMA into two v_add_f32 (it assumes VALU work is free there; on this part it is not, tools/mfma_peak.hip): 62 scalar
    // + 31 packed adds per two tiles became 62 packed ones, 222 -> 191 vector instructions per two tiles, 236 -> 206 VGPRs, -1 %.
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        f32x2 pa = pk_sub(IM_PAIR(sa, i), mm), pb = pk_sub(IM_PAIR(sb, i), mm);
        pa.x = __builtin_amdgcn_exp2f(pa.x); pa.y = __builtin_amdgcn_exp2f(pa.y);
        pb.x = __builtin_amdgcn_exp2f(pb.x); pb.y = __builtin_amdgcn_exp2f(pb.y);
        sa[2 * i] = pa.x; sa[2 * i + 1] = pa.y;
        sb[2 * i] = pb.x; sb[2 * i + 1] = pb.y;
        const f32x2 ps = pk_add(pa, pb);
        acc[i & 3] = i < 4 ? ps : pk_add(acc[i & 3], ps);
    }
    const f32x2 t2 = pk_add(pk_add(acc[0], acc[1]), pk_add(acc[2], acc[3]));
    __builtin_amdgcn_sched_barrier(0);
    float rs = t2.x + t2.y;
    rs += __shfl_xor(rs, 32);
    l_run += rs;
}

// O^T += V^T . P^T for one 64-key tile: register r of P is key acc_row(r, hh) of its half. V fragments are read
// from LDS in groups of four key rows, one group ahead of the MFMAs that consume them.
__device__ __forceinline__ void pv_tile(const float* __restrict__ sV, int c, int hh, const f32x16& pa, const f32x16& pb, f32x16& o0, f32x16& o1) {
#ifdef IM_ABL_PV_WIDE   // timing-only ablation: the instruction mix of a transposed V image (two 16-byte reads per 8 MFMAs)
    {
        const float* vt = sV + c * 68 + 4 * hh;      // stride 68: conflict-free 16-byte reads (what a transposed image would use)
        float4 a4 = *reinterpret_cast<const float4*>(vt), b4 = *reinterpret_cast<const float4*>(vt + 1900);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            float4 na4 = a4, nb4 = b4;
            if (g < 7) { na4 = *reinterpret_cast<const float4*>(vt + 8 * ((g + 1) & 3) + 32 * ((g + 1) >> 2)); nb4 = *reinterpret_cast<const float4*>(vt + 1900 + 8 * ((g + 1) & 3) + 32 * ((g + 1) >> 2)); }
            const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 4 * (g & 3) + q;
                const float p = (g < 4) ? pa[r] : pb[r];
                o0 = mfma32(av[q], p, o0);
                o1 = mfma32(bv[q], p, o1);
            }
            a4 = na4; b4 = nb4;
        }
        return;
    }
#endif
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

// G = key groups per block. G = 2: eight waves; waves 4..7 take the same 128 queries as waves 0..3 but the odd 64-key
// tiles, and the two partial (O, m, l) are merged through LDS at the end. At 4096 keypoints the grid is exactly one
// block per CU, so G = 2 is what puts two waves on every SIMD: one wave's softmax / LDS / barrier stalls are covered
// by the other's MFMAs.
template <int G>
__global__ __launch_bounds__(256 * G, 2 / G) void flash_attn_f32_kernel(AttnArgs a) {
    constexpr int NT = 256 * G;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    // XCD-aware block -> tile map: workgroups are dealt round-robin over the 8 XCDs (bid % 8), each with a private
    // 4 MB L2. The (image, head) index is the fastest-varying part of the linear block id, so with
    // heads * batch = 8 every query block of one (image, head) lands on the same XCD and that XCD's L2 holds exactly
    // one K/V set (2 MB at 4096 keys) instead of all eight.
    // With more than one pair in the batch the pair index is the SLOWEST part of the block id: the blocks resident at any
    // time belong to one or two pairs, so an XCD's L2 still holds one K/V set per (image, head) it serves.
    const int gsz = (a.batch % 2 == 0) ? 2 : a.batch;                  // images per group (a pair; or the whole odd batch)
    const int hz = a.heads * gsz;
    const int nqb = (a.n_max + 127) / 128;
    const int per_group = hz * nqb * (a.part ? ATTN_MAX_SPLIT : 1);
    const int grp_idx = blockIdx.x / per_group;
    const int bid = blockIdx.x - grp_idx * per_group;
    const int head = (bid % hz) % a.heads, z = grp_idx * gsz + (bid % hz) / a.heads;
    const int y = a.cross ? (z ^ 1) : z;
    if (a.active && a.active[(z >> 1) * a.pstride] == 0) return;
    const int nq = a.n_ptr ? a.n_ptr[(z >> 1) * a.pstride + (z & 1)] : a.n_max;
    const int nk_all = a.n_ptr ? a.n_ptr[(y >> 1) * a.pstride + (y & 1)] : a.n_max;
    const int qblk = (bid / hz) % nqb, split = (bid / hz) / nqb;
    const int qb = qblk * 128;
    if (qb >= nq || nk_all <= 0) return;
    // Split-KV: with few live queries (pruned pairs, small tiles) 128-query blocks leave most CUs idle, so the keys of a
    // query block are cut into n_split ranges of whole steps that run on separate blocks (the grid always holds
    // ATTN_MAX_SPLIT blocks per query block; the surplus exits here). Decided on the device: the live counts are not
    // known to the host.
    int n_split = 1, k0 = 0, nk = nk_all;
    if (a.part) {
        const int steps_all = (nk_all + KT * G - 1) / (KT * G);
        const int want = nq > 2048 ? 1 : (nq > 1024 ? 2 : ATTN_MAX_SPLIT);
        const int steps_per = (steps_all + want - 1) / want;
        n_split = (steps_all + steps_per - 1) / steps_per;
        k0 = split * steps_per * (KT * G);
        nk = min(nk_all - k0, steps_per * (KT * G));
    }
    if (split >= n_split) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, grp = tid >> 8;
    const int c = lane & 31, hh = lane >> 5;
    const int qrow = qb + wave * 32 + c;

    const float* Q = a.q + (long)z * a.bstride + (long)head * a.hstride;
    const __amdgpu_buffer_rsrc_t K = make_rsrc(a.k + (long)y * a.bstride + (long)head * a.hstride + (long)k0 * 64, nk);
    const __amdgpu_buffer_rsrc_t V = make_rsrc(a.v + (long)y * a.bstride + (long)head * a.hstride + (long)k0 * 64, nk);
    const unsigned voff = ((tid >> 4) * 64 + (tid & 15) * 4) * sizeof(float);

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
    const int nt = (nk + KT * G - 1) / (KT * G);   // steps of G tiles

    // ---- prologue: K0 -> LDS, S(0); K1, V0 -> LDS; K2, V1 in registers
    float4 rk0, rk1, rk2, rk3, rv0, rv1, rv2, rv3;
    constexpr int KSTG = G * KT * KS, VSTG = G * KT * VS;   // one ring stage = G tiles
    float* const sK0 = smem;
    float* const sV0 = smem + 2 * KSTG;
    const int gk = grp * KT * KS, gv = grp * KT * VS;       // this wave's tile inside a stage
    IM_LOAD_TILE(rk, K, 0)
    IM_STORE_TILE(rk, sK0, KS)
    IM_LOAD_TILE(rk, K, 1)
    IM_LOAD_TILE(rv, V, 0)
    __syncthreads();
    qk_tile(sK0 + gk, c, hh, qf, sa, sb);
    IM_STORE_TILE(rk, sK0 + KSTG, KS)
    IM_STORE_TILE(rv, sV0, VS)
    IM_LOAD_TILE(rk, K, 2)
    IM_LOAD_TILE(rv, V, 1)
    __syncthreads();

    // iteration t: stage K(t+2), V(t+1); prefetch K(t+3), V(t+2); QK^T of tile t+1; softmax of tile t; PV of tile t;
    // one barrier. Steps run two per trip with the score registers and the LDS ring stages swapped statically - no
    // register copies and no ring arithmetic; an odd leftover step takes the generic form, then the masked tail.
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
#define IM_STEP_FAST(KWS, KRS, VWS, VRS, SA, SB, NA, NB)                                \
    {                                                                                   \
        float* const kw = sK0 + (KWS) * KSTG;                                           \
        float* const vw = sV0 + (VWS) * VSTG;                                           \
        IM_ABL_STAGE(IM_STORE_TILE(rk, kw, KS)                                          \
        IM_STORE_TILE(rv, vw, VS)                                                       \
        IM_LOAD_TILE(rk, K, t + (KWS) + 3)                                              \
        IM_LOAD_TILE(rv, V, t + (KWS) + 2))                                             \
        qk_tile(sK0 + (KRS) * KSTG + gk, c, hh, qf, NA, NB);                            \
        softmax_tile<false>(SA, SB, 0, nk, hh, m_run, l_run, o0, o1);                   \
        pv_tile(sV0 + (VRS) * VSTG + gv, c, hh, SA, SB, o0, o1);                        \
        IM_ABL_BARRIER                                                                  \
    }
#define IM_STEP(TAIL)                                                                   \
    {                                                                                   \
        float* const kw = sK0 + (t & 1) * KSTG;                                         \
        float* const kr = sK0 + ((t + 1) & 1) * KSTG + gk;                              \
        float* const vw = sV0 + ((t + 1) & 1) * VSTG;                                   \
        float* const vr = sV0 + (t & 1) * VSTG + gv;                                    \
        if (!TAIL) {                                                                    \
            IM_ABL_STAGE(IM_STORE_TILE(rk, kw, KS)                                      \
            IM_STORE_TILE(rv, vw, VS)                                                   \
            IM_LOAD_TILE(rk, K, t + 3)                                                  \
            IM_LOAD_TILE(rv, V, t + 2))                                                 \
            qk_tile(kr, c, hh, qf, na, nb);                                             \
        }                                                                               \
        softmax_tile<TAIL>(sa, sb, t * (KT * G) + grp * KT, nk, hh, m_run, l_run, o0, o1); \
        pv_tile(vr, c, hh, sa, sb, o0, o1);                                             \
        if (!TAIL) { sa = na; sb = nb; }                                                \
        IM_ABL_BARRIER                                                                  \
    }
    int t = 0;
    for (; t + 2 <= nt - 1; t += 2) {            // KWS doubles as the step's parity: the ring stage K(t+2) goes to
        IM_STEP_FAST(0, 1, 1, 0, sa, sb, na, nb)
        IM_STEP_FAST(1, 0, 0, 1, na, nb, sa, sb)
    }
    for (; t < nt - 1; ++t) IM_STEP(false)
    IM_STEP(true)
#undef IM_STEP
#undef IM_STEP_FAST

    // ---- merge the key groups: group 1 parks (O, m, l) in LDS ([register][thread], conflict-free), group 0 folds it in
    if constexpr (G == 2) {
        float* const sc = smem;
        const int t1 = tid & 255;
        if (grp == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { sc[r * 256 + t1] = o0[r]; sc[(16 + r) * 256 + t1] = o1[r]; }
            sc[32 * 256 + t1] = m_run;
            sc[33 * 256 + t1] = l_run;
        }
        __syncthreads();
        if (grp == 1) return;
        const float m1 = sc[32 * 256 + t1], l1 = sc[33 * 256 + t1];
        const float m = fmaxf(m_run, m1);               // group 0 always saw key 0, so m is finite
        const float w0 = __builtin_amdgcn_exp2f(m_run - m), w1 = __builtin_amdgcn_exp2f(m1 - m);
        l_run = l_run * w0 + l1 * w1;
        m_run = m;                                      // (O, l) are now relative to m: the split-KV merge below needs it
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o0[r] = o0[r] * w0 + sc[r * 256 + t1] * w1;
            o1[r] = o1[r] * w0 + sc[(16 + r) * 256 + t1] * w1;
        }
    }

    // ---- split-KV: park the partial (O, m, l); the last block of this query block merges all of them in split order
    if (n_split > 1) {
        const long prow = (((long)z * a.heads + head) * a.n_max + min(qrow, a.n_max - 1)) * 66;
        const long pstride = (long)a.batch * a.heads * a.n_max * 66;
        float* pp = a.part + (long)split * pstride + prow;
        if (qrow < nq) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<float2*>(pp + 8 * g + 4 * hh) = make_float2(o0[4 * g], o0[4 * g + 1]);
            *reinterpret_cast<float2*>(pp + 8 * g + 4 * hh + 2) = make_float2(o0[4 * g + 2], o0[4 * g + 3]);
            *reinterpret_cast<float2*>(pp + 32 + 8 * g + 4 * hh) = make_float2(o1[4 * g], o1[4 * g + 1]);
            *reinterpret_cast<float2*>(pp + 32 + 8 * g + 4 * hh + 2) = make_float2(o1[4 * g + 2], o1[4 * g + 3]);
        }
        if (hh == 0) { pp[64] = m_run; pp[65] = l_run; }
        }
        __threadfence();
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem);
        if (tid == 0) {
            int* cnt = a.counters + ((long)z * a.heads + head) * nqb + qblk;
            const int old = atomicAdd(cnt, 1);
            *flag = (old == n_split - 1);
            if (old == n_split - 1) *cnt = 0;            // ready for the next launch
        }
        __syncthreads();
        if (!*flag) return;
        __threadfence();
        float m = -INFINITY;
        for (int sidx = 0; sidx < n_split; ++sidx) m = fmaxf(m, a.part[(long)sidx * pstride + prow + 64]);
        l_run = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
        for (int sidx = 0; sidx < n_split; ++sidx) {
            const float* ps = a.part + (long)sidx * pstride + prow;
            const float w = __builtin_amdgcn_exp2f(ps[64] - m);
            l_run += w * ps[65];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    o0[4 * g + q] += w * ps[8 * g + 4 * hh + q];
                    o1[4 * g + q] += w * ps[32 + 8 * g + 4 * hh + q];
                }
        }
    }

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

template <int G>
static hipError_t launch_g(const AttnArgs& a, hipStream_t s) {
    const size_t lds = G * ATTN_LDS_FLOATS * sizeof(float);
    static size_t lds_optin[IM_MAX_DEVICES] = {0};   // per device: a process may hold contexts on several GPUs
    if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(&flash_attn_f32_kernel<G>), lds, lds_optin); e != hipSuccess) return e;
    dim3 grid(((a.n_max + 127) / 128) * a.heads * a.batch * (a.part ? ATTN_MAX_SPLIT : 1)), block(256 * G);
    hipLaunchKernelGGL(flash_attn_f32_kernel<G>, grid, block, lds, s, a);
    return hipGetLastError();
}

hipError_t launch_flash_attn(const AttnArgs& a, hipStream_t s) {
    if (a.n_max <= 0) return hipSuccess;
    static const bool f32_form = [] { const char* e = getenv("IM_ATTN_F32"); return e && atoi(e) != 0; }();
    if (!f32_form && !a.f32_form) return launch_flash_attn_bx(a, s);
    static const int force = [] { const char* e = getenv("IM_ATTN_GROUPS"); return e ? atoi(e) : 0; }();
    const int g = force ? force : 2;
    return g == 2 ? launch_g<2>(a, s) : launch_g<1>(a, s);
}

}  // namespace im
