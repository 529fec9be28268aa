// fp32 "NT" GEMM on the f32-input matrix cores (v_mfma_f32_32x32x2_f32) with fused epilogues.
//
//   C_z[m][n] = epi( sum_k A_z[m][k] * W[n][k] )          A: [M][K] row-major, W: [N][K] row-major
//
// Used for every nn.Linear / Conv1d(k=1) / 1x1 Conv2d of the path (reference call sites:
// `lightglue/lightglue.py:153,160,162,192-193,212,277-281`, `lightglue/superpoint.py:169,204`,
// `SuperGlue/models/superglue.py:51-61,104-116,276-280`) and for the keypoint x keypoint score matrix
// (`lightglue/lightglue.py:280`, `superglue.py:279`).
//
// Tiling: 256 threads = 4 waves in a 2x2 grid; each wave owns (BM/2)x(BN/2) of the block tile as
// 32x32 MFMA tiles. K is consumed in slabs of BK = 32 (64-deep slabs measured neutral-to-slower at the LightGlue shapes) staged through LDS (row stride BK + 4 floats:
// the ds_read_b128 fragment reads are bank-conflict free). The contraction index inside a slab is
// permuted (lane-half h reads k = h BK/2 .. contiguously) so that one 16-byte LDS read feeds four
// MFMAs; A and B use the same permutation, so the product is unchanged up to summation order.
// Global loads of slab t+1 are issued before the MFMAs of slab t (register staging).
#include "common.h"
#include "kernels.h"
#include "sp_post.h"

namespace im {

typedef unsigned int gu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int gu32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 gbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gbf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x16 gmfma_bf(gu32x4 a, gu32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gbf16x8, a), __builtin_bit_cast(gbf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned gcvt_pk(float a, float b) {
    const gbf16x2 v = __builtin_convertvector(f32x2{a, b}, gbf16x2);
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void gsplit2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = gcvt_pk(a, b);
    float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = gcvt_pk(ra, rb);
    ra -= __uint_as_float(m << 16);
    rb -= __uint_as_float(m & 0xffff0000u);
    l = gcvt_pk(ra, rb);
}

// BX (round 5, GemmArgs::bx): the product on the bf16 matrix cores with fp32 accuracy, as attention_bx.hip - the fp32 tiles are cut into three
// bf16 planes as they are written to LDS (row stride 80 bytes: conflict-free 16-byte fragment reads), six v_mfma_f32_32x32x16_bf16 per 16 k
// and tile pair, small products first; everything around the slab loop (block map, staging loads, epilogues) is shared with the f32 form.
template <int BM, int BN, int BK, int EPI, bool BX>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs a) {
    constexpr int LDS_LD = BK + 4;
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int MB = WM / 32, NB = WN / 32;
    constexpr int F4 = BK / 4;                                 // float4 per row of a slab
    constexpr int A_IT = BM * F4 / 256, B_IT = BN * F4 / 256;  // float4 loads per thread per slab
    constexpr int XS = BK * 2 + 16, XPA = BM * XS, XPB = BN * XS;   // BX: plane row stride, plane sizes (bytes)
    __shared__ __attribute__((aligned(16))) float smem[BX ? 3 * (XPA + XPB) / 4 : (BM + BN) * LDS_LD];
    float* sA = smem;
    float* sB = smem + BM * LDS_LD;
    unsigned char* const xA = reinterpret_cast<unsigned char*>(smem);
    unsigned char* const xB = xA + 3 * XPA;

    // XCD-aware block -> tile map: the row-tile index is the fastest-varying part of the linear block id, so (with
    // a multiple of 8 row tiles) all column tiles of one row tile run on the same XCD and the A rows are fetched
    // into one L2 instead of up to eight.
    const int gm = (a.m_max + BM - 1) / BM;
    const int z = blockIdx.y;
    const int pair = a.pair_batched ? z : (z >> 1);
    if (a.active && a.active[pair * a.pstride] == 0) return;
    const int M = a.m_ptr ? a.m_ptr[a.pair_batched ? z * a.pstride : (z >> 1) * a.pstride + (z & 1)] : a.m_max;
    const int Nlive = a.n_ptr ? min(a.n_ptr[z * a.pstride], a.N) : a.N;
    const int m0 = (blockIdx.x % gm) * BM, n0 = (blockIdx.x / gm) * BN;
    if (m0 >= M || n0 >= Nlive) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;

    const int sel = a.sel ? a.sel[pair] : 0;
    const float* Wp = a.W + (long)z * a.w_bstride + (long)sel * a.w_sel_stride;
    const float* bias = a.bias ? a.bias + (long)sel * a.bias_sel_stride : nullptr;
    const float* A0 = a.A + (long)z * a.a_bstride;
    const float* A1 = a.A1 ? a.A1 + (long)z * a.a1_bstride : nullptr;

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Register staging of the next slab in named registers (arrays indexed inside helper lambdas / loops ended up in
    // scratch). Buffer loads: descriptors sized to the live rows (rows beyond them read as zero; their products are never
    // stored), lane offsets fixed for the whole kernel, the slab is a scalar offset - no address arithmetic on the VALU,
    // which on gfx950 would cost matrix-pipe time (tools/mfma_peak.hip).
    static_assert((A_IT == 1 || A_IT == 2 || A_IT == 4) && (B_IT == 1 || B_IT == 2 || B_IT == 4), "staging registers");
    const __amdgpu_buffer_rsrc_t rA0 = gmake_rsrc(A0, (unsigned)M * a.lda * 4u);
    const __amdgpu_buffer_rsrc_t rA1 = gmake_rsrc(A1 ? A1 : A0, A1 ? (unsigned)M * a.lda1 * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rW = gmake_rsrc(Wp, (unsigned)Nlive * a.ldw * 4u);
    unsigned va0[A_IT], va1[A_IT], vb[B_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const unsigned row = m0 + (tid + i * 256) / F4, col = ((tid + i * 256) % F4) * 4;
        va0[i] = (row * a.lda + col) * 4u;
        va1[i] = (row * a.lda1 + col) * 4u;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) vb[i] = ((unsigned)(n0 + (tid + i * 256) / F4) * a.ldw + ((tid + i * 256) % F4) * 4) * 4u;
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    ra1 = ra2 = ra3 = rb1 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
#define IM_LD_A(i) gbuf_load4(second_ ? rA1 : rA0, second_ ? va1[i] : va0[i], so_)
#define IM_LD_B(i, k0_) gbuf_load4(rW, vb[i], (k0_) * 4u)
#define IM_LOAD_SLAB(k0_)                                                                  \
    {                                                                                      \
        const bool second_ = A1 && (k0_) >= a.ksplit;                                      \
        const unsigned so_ = (second_ ? (k0_) - a.ksplit : (k0_)) * 4u;                    \
        ra0 = IM_LD_A(0);                                                                  \
        if constexpr (A_IT > 1) ra1 = IM_LD_A(1);                                          \
        if constexpr (A_IT > 2) { ra2 = IM_LD_A(2); ra3 = IM_LD_A(3); }                    \
        rb0 = IM_LD_B(0, k0_);                                                             \
        if constexpr (B_IT > 1) rb1 = IM_LD_B(1, k0_);                                     \
        if constexpr (B_IT > 2) { rb2 = IM_LD_B(2, k0_); rb3 = IM_LD_B(3, k0_); }          \
    }
#define IM_ST_F32(buf, i, r) *reinterpret_cast<float4*>(buf + ((tid + (i) * 256) / F4) * LDS_LD + ((tid + (i) * 256) % F4) * 4) = r
    auto st_bx = [&](unsigned char* plane0, int plane_bytes, int i, float4 x) {
        unsigned h0, m0, l0, h1, m1, l1;
#if defined(IM_GABL_NO_BCUT) || defined(IM_GABL_NO_CUT)      // timing ablations (bf16-accurate results): what the plane cut of W / of both operands costs
#ifdef IM_GABL_NO_CUT
        const bool skip_ = true;
#else
        const bool skip_ = plane0 == xB;
#endif
        if (skip_) { h0 = gcvt_pk(x.x, x.y); h1 = gcvt_pk(x.z, x.w); m0 = l0 = m1 = l1 = 0u; }     // bf16-accurate values: the data-dependent flow stays
        else
#endif
        {
        gsplit2(x.x, x.y, h0, m0, l0);
        gsplit2(x.z, x.w, h1, m1, l1);
        }
        unsigned char* d = plane0 + ((tid + i * 256) / F4) * XS + ((tid + i * 256) % F4) * 8;
        *reinterpret_cast<gu32x2*>(d) = gu32x2{h0, h1};
        *reinterpret_cast<gu32x2*>(d + plane_bytes) = gu32x2{m0, m1};
        *reinterpret_cast<gu32x2*>(d + 2 * plane_bytes) = gu32x2{l0, l1};
    };
#define IM_ST(buf, i, r)                                                              \
    {                                                                                 \
        if constexpr (BX) st_bx(buf == sA ? xA : xB, buf == sA ? XPA : XPB, i, r);    \
        else IM_ST_F32(buf, i, r);                                                    \
    }

    IM_LOAD_SLAB(0)
    for (int k0 = 0; k0 < a.K; k0 += BK) {
        IM_ST(sA, 0, ra0);
        if constexpr (A_IT > 1) IM_ST(sA, 1, ra1);
        if constexpr (A_IT > 2) { IM_ST(sA, 2, ra2); IM_ST(sA, 3, ra3); }
        IM_ST(sB, 0, rb0);
        if constexpr (B_IT > 1) IM_ST(sB, 1, rb1);
        if constexpr (B_IT > 2) { IM_ST(sB, 2, rb2); IM_ST(sB, 3, rb3); }
        __syncthreads();
        if (k0 + BK < a.K) IM_LOAD_SLAB(k0 + BK)
        if constexpr (BX) {
#pragma unroll
            for (int kc = 0; kc < BK / 16; ++kc) {
                gu32x4 fa[MB][3], fb[NB][3];
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        fa[i][pl] = *reinterpret_cast<const gu32x4*>(xA + pl * XPA + (wm0 + i * 32 + c) * XS + kc * 32 + hh * 16);
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        fb[j][pl] = *reinterpret_cast<const gu32x4*>(xB + pl * XPB + (wn0 + j * 32 + c) * XS + kc * 32 + hh * 16);
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        acc[i][j] = gmfma_bf(fa[i][0], fb[j][2], acc[i][j]);
                        acc[i][j] = gmfma_bf(fa[i][2], fb[j][0], acc[i][j]);
                        acc[i][j] = gmfma_bf(fa[i][1], fb[j][1], acc[i][j]);
                        acc[i][j] = gmfma_bf(fa[i][0], fb[j][1], acc[i][j]);
                        acc[i][j] = gmfma_bf(fa[i][1], fb[j][0], acc[i][j]);
                        acc[i][j] = gmfma_bf(fa[i][0], fb[j][0], acc[i][j]);
                    }
            }
        } else
#pragma unroll
        for (int t = 0; t < BK / 8; ++t) {
            float4 fa[MB], fb[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i)
                fa[i] = *reinterpret_cast<const float4*>(sA + (wm0 + i * 32 + c) * LDS_LD + hh * (BK / 2) + t * 4);
#pragma unroll
            for (int j = 0; j < NB; ++j)
                fb[j] = *reinterpret_cast<const float4*>(sB + (wn0 + j * 32 + c) * LDS_LD + hh * (BK / 2) + t * 4);
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    acc[i][j] = mfma32(fa[i].x, fb[j].x, acc[i][j]);
                    acc[i][j] = mfma32(fa[i].y, fb[j].y, acc[i][j]);
                    acc[i][j] = mfma32(fa[i].z, fb[j].z, acc[i][j]);
                    acc[i][j] = mfma32(fa[i].w, fb[j].w, acc[i][j]);
                }
        }
        __syncthreads();
    }
#undef IM_LOAD_SLAB
#undef IM_LD_A
#undef IM_LD_B
#undef IM_ST
#undef IM_ST_F32

    // ---- epilogue: lane holds column n = .. + c, rows row0 + 4 hh + (r & 3) + 8 (r >> 2). Buffer stores: the descriptor
    // covers the live rows only, so rows past them are dropped by the range check; a dead column gets an out-of-range
    // lane offset; the row of register r is a scalar offset. No per-element branch, no 64-bit address arithmetic.
    constexpr unsigned OOB = 0xFFFFF000u;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int colbase = n0 + wn0 + j * 32;            // wave-uniform
            const int col = colbase + c;
            const bool col_ok = col < Nlive;
            const float bv = (bias && col_ok) ? bias[col] : 0.f;
            const unsigned rowl = m0 + wm0 + i * 32 + 4 * hh; // row of register 0 of this lane
            if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_RESID) {
                const __amdgpu_buffer_rsrc_t rC = gmake_rsrc(a.C + (long)z * a.c_bstride, (unsigned)M * a.ldc * 4u);
                const unsigned vo = col_ok ? (rowl * a.ldc + col) * 4u : OOB;
                const unsigned rstep = (unsigned)a.ldc * 4u;
                float rv[16];
                if constexpr (EPI == EPI_BIAS_RESID) {
                    const __amdgpu_buffer_rsrc_t rR = gmake_rsrc(a.R + (long)z * a.r_bstride, (unsigned)M * a.ldr * 4u);
                    const unsigned vr = col_ok ? (rowl * a.ldr + col) * 4u : OOB;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        rv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rR, vr, ((r & 3) + 8 * (r >> 2)) * (unsigned)a.ldr * 4u, 0));
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float val = acc[i][j][r] + bv;
                    if constexpr (EPI == EPI_BIAS) val = a.alpha * val;
                    else if constexpr (EPI == EPI_BIAS_RELU) val = fmaxf(val, 0.f);
                    else val = rv[r] + val;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), rC, vo, ((r & 3) + 8 * (r >> 2)) * rstep, 0);
                }
            } else {
                // head-major stores [head][row][64]; which / head are wave-uniform (a tile spans 32 of a head's 64 columns)
                int which = 0, hd = colbase;
                if constexpr (EPI == EPI_QKV_ROPE || EPI == EPI_HEADS_QV) { which = colbase >> 8; hd = colbase & 255; }
                float* dst = a.q;
                if constexpr (EPI == EPI_QKV_ROPE) dst = which == 0 ? a.q : (which == 1 ? a.k : a.v);
                if constexpr (EPI == EPI_HEADS_QV) dst = which ? a.v : a.q;
                const __amdgpu_buffer_rsrc_t rD = gmake_rsrc(dst + (long)z * a.head_bstride + (long)(hd >> 6) * a.head_stride, (unsigned)M * 256u);
                const int d = (hd & 63) + c;
                const unsigned vo = col_ok ? (rowl * 64 + d) * 4u : OOB;
                float scale = a.alpha;
                if constexpr (EPI == EPI_QKV_ROPE) scale = 1.f;
                if constexpr (EPI == EPI_HEADS_QV) scale = which ? 1.f : a.alpha;
                const bool rope = EPI == EPI_QKV_ROPE && which < 2 && a.cs;
                if (rope) {
                    // rotary (`lightglue/lightglue.py:49-57`): pairs (2i, 2i+1) share a frequency;
                    // out = t * cos + rotate_half(t) * sin, rotate_half: (x0, x1) -> (-x1, x0)
                    const __amdgpu_buffer_rsrc_t rCs = gmake_rsrc(a.cs + (long)z * a.enc_bstride, (unsigned)M * 128u);
                    const __amdgpu_buffer_rsrc_t rSn = gmake_rsrc(a.sn + (long)z * a.enc_bstride, (unsigned)M * 128u);
                    const unsigned ve = (rowl * 32 + (d >> 1)) * 4u;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned so = ((r & 3) + 8 * (r >> 2)) * 128u;
                        const float cs = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rCs, ve, so, 0));
                        const float sn = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rSn, ve, so, 0));
                        const float val = acc[i][j][r] + bv;
                        const float partner = __shfl_xor(val, 1);
                        const float outv = (d & 1) ? (val * cs) + (partner * sn) : (val * cs) + ((-partner) * sn);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(outv), rD, vo, ((r & 3) + 8 * (r >> 2)) * 256u, 0);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(scale * (acc[i][j][r] + bv)), rD, vo, ((r & 3) + 8 * (r >> 2)) * 256u, 0);
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// The K = 256 projections of a transformer block as ROW BLOCKS (round 6; `lightglue/lightglue.py:153, 192-193`: Wqkv + rotary, to_qk / to_v; SuperGlue's
// `proj`: `superglue.py:79-93`). The tiled kernel above runs them at 0.27 of the bf16 / 6 roofline (281 us per ten pairs for qkv): eight 32-deep slabs, each with
// its own staging, plane cut of BOTH operands (the W tile again in every row block, the x tile again in every column block) and barrier, around 48 MFMAs per
// wave. Here a block owns 32 rows for ALL N output columns, as the fused feed-forward does: x is cut ONCE into three 256-wide bf16 planes in LDS (50 KB: up to
// three blocks per CU), the weights arrive cut (the host's pack_frag_weights: MFMA-fragment order, a wave's 16-byte-per-lane load is one contiguous KiB) and
// stream from L2 straight into registers THREE 16-deep steps ahead (four register buffers; with one step ahead a step of six MFMAs waited 830 cycles for its
// weights), one barrier per block. Wave w owns the column tiles w, w + 8, (w + 16): one of q / k / v each, sixteen steps of six MFMAs per tile, then that tile's
// epilogue (the tiled kernel's, statement for statement: bias, rotary on q / k, head-major stores). The products and their order per accumulator are those of
// the tiled kernel (x planes as A, W planes as B, k ascending, h l | l h | m m | h m | m h | h h): bit-identical outputs (profiles/r06_proj_rows.txt).
namespace pr {
constexpr int BM = 32, NT = 512;
constexpr int PS = 528;              // bytes per row of a plane: 256 x 2 + 16 (132 dwords = 4 mod 64: conflict-free 16-byte row reads)
constexpr int PLANE = BM * PS;       // 16,896
constexpr int LDS_BYTES = 3 * PLANE; // 50,688
constexpr unsigned TILE_BYTES = (256 / 16) * 3 * 1024;   // one packed 32-column tile at K = 256: 16 k chunks x 3 planes x 1 KiB
}  // namespace pr

#ifdef IM_PSTAMP   // diagnostic build only (tools/proj_stamps.py): shader-clock stamps of wave 0 of every block, never read by the kernel
__device__ unsigned long long g_pstamp[4096 * 16];
#define P_STAMP(i) { if (wave == 0) { const unsigned long long ts_ = __builtin_amdgcn_s_memtime(); const int b_ = blockIdx.y * gridDim.x + blockIdx.x; if (lane == 0 && b_ < 4096) g_pstamp[b_ * 16 + (i)] = ts_; } }
extern "C" int im_debug_pstamps(unsigned long long* host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pstamp), n * sizeof(unsigned long long));
}
#else
#define P_STAMP(i)
#endif

template <int EPI, int TILES>
__global__ __launch_bounds__(pr::NT, 4) void proj_rows_kernel(GemmArgs a) {
    using namespace pr;
    static_assert(EPI == EPI_QKV_ROPE || EPI == EPI_HEADS_QV, "the two projections of a LightGlue block");
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
    const int z = blockIdx.y, pair = z >> 1;
    if (a.active && a.active[pair * a.pstride] == 0) return;
    const int M = a.m_ptr ? a.m_ptr[pair * a.pstride + (z & 1)] : a.m_max;
    const int m0 = blockIdx.x * BM;
    if (m0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, hh = lane >> 5;
    const __amdgpu_buffer_rsrc_t rX = gmake_rsrc(a.A + (long)z * a.a_bstride, (unsigned)M * 1024u);
    const __amdgpu_buffer_rsrc_t rW = gmake_rsrc(a.wp, (unsigned)a.N * 256u * 6u);

    P_STAMP(0)
    // weight steps (one 16-deep k chunk of one column tile = three 1 KB loads) run THREE steps ahead through four register buffers: the step index s = 16 j + kc
    // walks this wave's tiles j * 8 + wave; steps past the last tile are out of the descriptor's range by their lane offset (zeros, no traffic, no branch)
    gu32x4 b0[3], b1[3], b2[3], b3[3];
#define PR_LOAD(b_, s_)                                                                                                          \
    {                                                                                                                            \
        const bool in_ = (s_) < TILES * 16;                                                                                      \
        const unsigned so_ = in_ ? (unsigned)((((s_) >> 4) * 8 + wave)) * TILE_BYTES + (unsigned)(((s_) & 15) * 3) * 1024u : 0u;   \
        const unsigned vo_ = in_ ? lane * 16u : 0x80000000u;    /* out of range by the LANE offset: a scalar offset beyond num_records would wrap the range check */ \
        _Pragma("unroll") for (int g = 0; g < 3; ++g) b_[g] = __builtin_amdgcn_raw_buffer_load_b128(rW, vo_, so_ + (unsigned)g * 1024u, 0); \
    }
    PR_LOAD(b0, 0)
    PR_LOAD(b1, 1)
    PR_LOAD(b2, 2)
    {   // 32 rows x 256 floats -> three planes: thread -> (row = idx >> 6, float4 idx & 63), 4 float4 per thread
        float4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * NT;
            v[i] = gbuf_load4(rX, (unsigned)((m0 + (idx >> 6)) * 256 + (idx & 63) * 4) * 4u, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * NT;
            unsigned h0, m0_, l0, h1, m1, l1;
            gsplit2(v[i].x, v[i].y, h0, m0_, l0);
            gsplit2(v[i].z, v[i].w, h1, m1, l1);
            unsigned char* d = psm + (idx >> 6) * PS + (idx & 63) * 8;
            *reinterpret_cast<gu32x2*>(d) = gu32x2{h0, h1};
            *reinterpret_cast<gu32x2*>(d + PLANE) = gu32x2{m0_, m1};
            *reinterpret_cast<gu32x2*>(d + 2 * PLANE) = gu32x2{l0, l1};
        }
    }
    __syncthreads();
    P_STAMP(1)

#define PR_SIX(b_, kc_)                                                                                              \
    {                                                                                                                \
        const unsigned char* ap = psm + c * PS + ((kc_) * 16 + hh * 8) * 2;                                          \
        const gu32x4 ah = *reinterpret_cast<const gu32x4*>(ap), am = *reinterpret_cast<const gu32x4*>(ap + PLANE),   \
                     al = *reinterpret_cast<const gu32x4*>(ap + 2 * PLANE);                                          \
        acc = gmfma_bf(ah, b_[2], acc);                                                                              \
        acc = gmfma_bf(al, b_[0], acc);                                                                              \
        acc = gmfma_bf(am, b_[1], acc);                                                                              \
        acc = gmfma_bf(ah, b_[1], acc);                                                                              \
        acc = gmfma_bf(am, b_[0], acc);                                                                              \
        acc = gmfma_bf(ah, b_[0], acc);                                                                              \
    }
#pragma unroll 1
    for (int j = 0; j < TILES; ++j) {
        const int tile = j * 8 + wave;                       // wave-uniform: columns [32 tile, 32 tile + 32)
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 1
        for (int kc = 0; kc < 16; kc += 4) {
            const int s_ = j * 16 + kc;
            // a scheduling fence per step: left alone the machine scheduler sinks every load to its use (one step of latency hiding instead of three)
            PR_LOAD(b3, s_ + 3)
            __builtin_amdgcn_sched_barrier(0);
            PR_SIX(b0, kc)
            __builtin_amdgcn_sched_barrier(0);
            PR_LOAD(b0, s_ + 4)
            __builtin_amdgcn_sched_barrier(0);
            PR_SIX(b1, kc + 1)
            __builtin_amdgcn_sched_barrier(0);
            PR_LOAD(b1, s_ + 5)
            __builtin_amdgcn_sched_barrier(0);
            PR_SIX(b2, kc + 2)
            __builtin_amdgcn_sched_barrier(0);
            PR_LOAD(b2, s_ + 6)
            __builtin_amdgcn_sched_barrier(0);
            PR_SIX(b3, kc + 3)
            __builtin_amdgcn_sched_barrier(0);
        }
        P_STAMP(2 + 2 * j)
        // ---- epilogue of this tile (gemm_nt_kernel's head-major branch): lane holds column colbase + c, rows m0 + 4 hh + (r & 3) + 8 (r >> 2)
        const int colbase = tile * 32;
        const int col = colbase + c;
        const float bv = a.bias ? a.bias[col] : 0.f;
        const unsigned rowl = m0 + 4 * hh;
        const int which = colbase >> 8, hd = colbase & 255;
        float* dst = a.q;
        if constexpr (EPI == EPI_QKV_ROPE) dst = which == 0 ? a.q : (which == 1 ? a.k : a.v);
        if constexpr (EPI == EPI_HEADS_QV) dst = which ? a.v : a.q;
        const __amdgpu_buffer_rsrc_t rD = gmake_rsrc(dst + (long)z * a.head_bstride + (long)(hd >> 6) * a.head_stride, (unsigned)M * 256u);
        const int d = (hd & 63) + c;
        const unsigned vo = (rowl * 64 + d) * 4u;
        float scale = a.alpha;
        if constexpr (EPI == EPI_QKV_ROPE) scale = 1.f;
        if constexpr (EPI == EPI_HEADS_QV) scale = which ? 1.f : a.alpha;
        const bool rope = EPI == EPI_QKV_ROPE && which < 2 && a.cs;
        if (rope) {
            const __amdgpu_buffer_rsrc_t rCs = gmake_rsrc(a.cs + (long)z * a.enc_bstride, (unsigned)M * 128u);
            const __amdgpu_buffer_rsrc_t rSn = gmake_rsrc(a.sn + (long)z * a.enc_bstride, (unsigned)M * 128u);
            const unsigned ve = (rowl * 32 + (d >> 1)) * 4u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned so = ((r & 3) + 8 * (r >> 2)) * 128u;
                const float cs = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rCs, ve, so, 0));
                const float sn = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rSn, ve, so, 0));
                const float val = acc[r] + bv;
                const float partner = __shfl_xor(val, 1);
                const float outv = (d & 1) ? (val * cs) + (partner * sn) : (val * cs) + ((-partner) * sn);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(outv), rD, vo, ((r & 3) + 8 * (r >> 2)) * 256u, 0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(scale * (acc[r] + bv)), rD, vo, ((r & 3) + 8 * (r >> 2)) * 256u, 0);
        }
        P_STAMP(3 + 2 * j)
    }
#undef PR_LOAD
#undef PR_SIX
}

template <int EPI, int TILES>
static hipError_t launch_proj_rows_t(const GemmArgs& a, hipStream_t s) {
    static size_t lds_optin[IM_MAX_DEVICES] = {0};
    if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(&proj_rows_kernel<EPI, TILES>), pr::LDS_BYTES, lds_optin); e != hipSuccess) return e;
    const dim3 grid((a.m_max + pr::BM - 1) / pr::BM, a.batch), block(pr::NT);
    hipLaunchKernelGGL((proj_rows_kernel<EPI, TILES>), grid, block, pr::LDS_BYTES, s, a);
    return hipGetLastError();
}

// GemmArgs::wp set (W's planes in fragment order, pack_frag_weights(W, N, 256)): K = 256, lda = 256, N = 768 with EPI_QKV_ROPE or N = 512 with EPI_HEADS_QV,
// no second A, no per-batch / selected weights, every column live
hipError_t launch_proj_rows(const GemmArgs& a, hipStream_t s) {
    if (!a.wp || a.K != 256 || a.lda != 256 || a.A1 || a.sel || a.w_bstride || a.n_ptr || a.pair_batched) return hipErrorInvalidValue;
    if (a.m_max <= 0 || a.batch <= 0) return hipSuccess;
    if (a.epi == EPI_QKV_ROPE && a.N == 768) return launch_proj_rows_t<EPI_QKV_ROPE, 3>(a, s);
    if (a.epi == EPI_HEADS_QV && a.N == 512) return launch_proj_rows_t<EPI_HEADS_QV, 2>(a, s);
    return hipErrorInvalidValue;
}

hipError_t launch_gemm(const GemmArgs& a, hipStream_t s) {
    if (a.K % 32 != 0 || (a.A1 && a.ksplit % 64 != 0)) return hipErrorInvalidValue;
    const int bm = a.big_tile ? 128 : 64, bn = bm;
    dim3 grid(((a.N + bn - 1) / bn) * ((a.m_max + bm - 1) / bm), a.batch), block(256);
    if (grid.x == 0) return hipSuccess;
#define IM_GEMM_CASE(E)                                                                  \
    case E:                                                                              \
        if (a.bx) {                                                                      \
            if (a.big_tile) hipLaunchKernelGGL((gemm_nt_kernel<128, 128, 32, E, true>), grid, block, 0, s, a);   \
            else hipLaunchKernelGGL((gemm_nt_kernel<64, 64, 32, E, true>), grid, block, 0, s, a);               \
        } else {                                                                         \
            if (a.big_tile) hipLaunchKernelGGL((gemm_nt_kernel<128, 128, 32, E, false>), grid, block, 0, s, a);  \
            else hipLaunchKernelGGL((gemm_nt_kernel<64, 64, 32, E, false>), grid, block, 0, s, a);              \
        }                                                                                \
        break;
    switch (a.epi) {
        IM_GEMM_CASE(EPI_BIAS)
        IM_GEMM_CASE(EPI_BIAS_RESID)
        IM_GEMM_CASE(EPI_HEADS)
        IM_GEMM_CASE(EPI_QKV_ROPE)
        IM_GEMM_CASE(EPI_BIAS_RELU)
        IM_GEMM_CASE(EPI_HEADS_QV)
        default: return hipErrorInvalidValue;
    }
#undef IM_GEMM_CASE
    return hipGetLastError();
}

}  // namespace im
