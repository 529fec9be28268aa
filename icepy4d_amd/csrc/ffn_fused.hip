// LightGlue's feed-forward tail of a transformer block as ONE kernel (`lightglue/lightglue.py:144-149, 160-162, 212-216`):
//
//   x += W3 . gelu(layernorm(W0 . [x | att] + b0)) + b3          W0: 512 x 512 (out_proj / to_out folded in), W3: 256 x 512
//
// It replaces three launches (ffn.0 GEMM, LayerNorm + GELU, ffn.3 GEMM + residual) and the two HBM round trips of the 512-wide
// hidden rows (3 x 16.8 MB per launch at 2 x 4096 keypoints).
//
// Round 5: both products run on the bf16 matrix cores with fp32 accuracy, as the attention does (attention_bx.hip: an fp32 value is the
// exact sum of three bf16 values, six bf16 products per fp32 product, fp32 accumulation; error at or below the f32-input MFMA chain's,
// profiles/r05_bf16x_probe.txt). The weights are cut into their three planes on the host, once per weight set; the rows are cut when
// they are written to LDS - the inputs by the staging pass, the hidden rows by the LayerNorm / GELU pass - so every value is cut once.
//
// A block owns 32 rows of one image and holds them in LDS for the whole kernel: first [x | att] as three bf16 planes (32 x 512 each),
// then - over the same bytes - the hidden rows in fp32 for the normalisation, then their planes. 8 waves: in the first GEMM a wave owns 64
// of the 512 hidden columns (two 32 x 32 MFMA tiles), in the second 32 of the 256 output columns (one tile). With a single row tile
// per block nothing of W is shared between the waves of a block, so W never goes through LDS: the planes are packed in MFMA-fragment
// order ([column tile][k chunk of 16][plane][lane][8 bf16]: a wave's 16-byte-per-lane load is one contiguous KiB) and stream from L2
// straight into registers, double-buffered one 32-deep k chunk (24 / 12 MFMAs per wave) ahead.
//
// LayerNorm is the two-pass form of lg_misc.hip's layernorm_gelu_kernel (mean, then centred sum of squares, eps 1e-5, erf GELU).
#include "common.h"
#include "kernels.h"
#include "sp_post.h"

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace im {

namespace ff {
constexpr int BM = 32;              // rows per block
constexpr int NT = 512;             // threads per block (8 waves)
constexpr int LD = 516;             // fp32 row stride of the hidden rows (floats)
constexpr int PS = 1040;            // bytes per row of a bf16 plane: 512 x 2 + 16 (conflict-free 16-byte row reads)
constexpr int PLANE = BM * PS;      // 33,280 bytes
constexpr int LDS_BYTES = 3 * PLANE;                    // 99,840 (>= BM * LD * 4 = 66,048)
constexpr unsigned TILE_BYTES = (512 / 16) * 3 * 1024;  // one packed 32-column tile at K = 512: 32 k chunks x 3 planes x 1 KiB
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
}  // namespace ff

// host: fp32 -> bf16, round to nearest even (what v_cvt_pk_bf16_f32 does); the weights are finite
static inline uint16_t host_bf16(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float host_bf16_to_float(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// Fragment-order packing of a row-major W[N][K] as three bf16 planes (host side, once per weight set):
//   out[tile = n / 32][chunk = k / 16][plane][lane = hh * 32 + c][j] = plane of W[tile * 32 + c][16 chunk + 8 hh + j],  j = 0..7
// (the B operand of v_mfma_f32_32x32x16_bf16: lane (c, hh) holds k = 8 hh + j of column c). 6 bytes per weight: the vector holds
// n k 3 / 2 floats' worth of bytes.
std::vector<float> pack_frag_weights(const float* w, int n, int k) {
    std::vector<float> out((size_t)n * k * 3 / 2);
    uint16_t* o = reinterpret_cast<uint16_t*>(out.data());
    const int chunks = k / 16;
    for (int t = 0; t < n / 32; ++t)
        for (int ch = 0; ch < chunks; ++ch)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const float x = w[(size_t)(t * 32 + (lane & 31)) * k + 16 * ch + 8 * (lane >> 5) + j];
                    const uint16_t h = host_bf16(x);
                    const float r1 = x - host_bf16_to_float(h);
                    const uint16_t m = host_bf16(r1);
                    const float r2 = r1 - host_bf16_to_float(m);
                    const uint16_t l = host_bf16(r2);
                    const size_t base = (((size_t)t * chunks + ch) * 3) * 512 + (size_t)lane * 8 + j;     // in bf16 elements; a plane block = 64 x 8
                    o[base] = h; o[base + 512] = m; o[base + 1024] = l;
                }
    return out;
}

namespace ff {
__device__ __forceinline__ f32x16 mfma_bf(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
    const bf16x2 v = __builtin_convertvector(f32x2{a, b}, bf16x2);
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pk(a, b);
    float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk(ra, rb);
    ra -= __uint_as_float(m << 16);
    rb -= __uint_as_float(m & 0xffff0000u);
    l = cvt_pk(ra, rb);
}
// four consecutive row values -> 8 bytes in each plane
template <int PLANE_BYTES = PLANE>
__device__ __forceinline__ void put4(unsigned char* plane0, int off, float4 x) {
    unsigned h0, m0, l0, h1, m1, l1;
    split2(x.x, x.y, h0, m0, l0);
    split2(x.z, x.w, h1, m1, l1);
    *reinterpret_cast<u32x2*>(plane0 + off) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(plane0 + PLANE_BYTES + off) = u32x2{m0, m1};
    *reinterpret_cast<u32x2*>(plane0 + 2 * PLANE_BYTES + off) = u32x2{l0, l1};
}
// the two-blocks-per-CU form (ffn_fused_split_kernel): planes of HALF the contraction (256 of 512) at a time
constexpr int PS2 = 528;             // bytes per row of a half-K plane: 256 x 2 + 16 (132 dwords = 4 mod 64, as PS: conflict-free 16-byte row reads)
constexpr int PLANE2 = BM * PS2;     // 16,896 bytes; three of them (50,688) lie over the fp32 hidden rows
constexpr int LDS_BYTES2 = BM * LD * 4;   // 66,048: the 32 hidden rows in fp32
}  // namespace ff

#ifdef IM_FSTAMP   // diagnostic build only (tools/ffn_stamps.py): shader-clock stamps of wave 0 of every block, never read by the kernel
__device__ unsigned long long g_fstamp[4096 * 16];
#define F_STAMP(i) { if (wave == 0) { const unsigned long long ts_ = __builtin_amdgcn_s_memtime(); const int b_ = blockIdx.y * gridDim.x + blockIdx.x; if (lane == 0 && b_ < 4096) g_fstamp[b_ * 16 + (i)] = ts_; } }
extern "C" int im_debug_fstamps(unsigned long long* host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fstamp), n * sizeof(unsigned long long));
}
#else
#define F_STAMP(i)
#endif

// ACT 0: LayerNorm(512) + GELU between the two products (LightGlue); ACT 1: ReLU (SuperGlue's MLP with its BatchNorm folded into
// W0 / b0, `SuperGlue/models/superglue.py:51-61, 104-116`)
template <int ACT>
__global__ __launch_bounds__(ff::NT, 2) void ffn_fused_kernel(FfnArgs a) {
    using namespace ff;
    extern __shared__ __attribute__((aligned(16))) unsigned char sB[];
    float* const sA = reinterpret_cast<float*>(sB);
    const int z = blockIdx.y, pair = z >> 1;
    if (a.active && a.active[pair * a.pstride] == 0) return;
    const int M = a.m_ptr ? a.m_ptr[pair * a.pstride + (z & 1)] : a.m_max;
    const int m0 = blockIdx.x * BM;
    if (m0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // in an SGPR: everything derived from it is wave-uniform
    const int c = lane & 31, hh = lane >> 5;

    const __amdgpu_buffer_rsrc_t rX = gmake_rsrc(a.x + (long)z * a.x_bstride, (unsigned)M * 1024u);
    const __amdgpu_buffer_rsrc_t rAtt = gmake_rsrc(a.att + (long)z * a.att_bstride, (unsigned)M * 1024u);
    const __amdgpu_buffer_rsrc_t rW0 = gmake_rsrc(a.w0p, 512u * 512u * 6u);
    const __amdgpu_buffer_rsrc_t rW3 = gmake_rsrc(a.w3p, 256u * 512u * 6u);

    // a weight chunk = 32 k = two MFMA k chunks x three planes (x two column tiles in the first product): 16-byte loads, 1 KiB per wave each
    const unsigned vb0 = (unsigned)(2 * wave) * TILE_BYTES + lane * 16u;
    u32x4 p0[6], p1[6], q0[6], q1[6];
#define FF_LOAD2(b0_, b1_, ch_)                                                                                  \
    _Pragma("unroll") for (int g = 0; g < 6; ++g) {                                                              \
        b0_[g] = __builtin_amdgcn_raw_buffer_load_b128(rW0, vb0, (unsigned)((ch_) * 6 + g) * 1024u, 0);              \
        b1_[g] = __builtin_amdgcn_raw_buffer_load_b128(rW0, vb0 + TILE_BYTES, (unsigned)((ch_) * 6 + g) * 1024u, 0); \
    }
    // the six products of one 16-deep k chunk, small ones first: A planes (h, m, l) from LDS, B planes b[3 kc + (0, 1, 2)]
#define FF_SIX(acc_, ah_, am_, al_, b_, kc_)              \
    acc_ = mfma_bf(ah_, b_[3 * (kc_) + 2], acc_);         \
    acc_ = mfma_bf(al_, b_[3 * (kc_) + 0], acc_);         \
    acc_ = mfma_bf(am_, b_[3 * (kc_) + 1], acc_);         \
    acc_ = mfma_bf(ah_, b_[3 * (kc_) + 1], acc_);         \
    acc_ = mfma_bf(am_, b_[3 * (kc_) + 0], acc_);         \
    acc_ = mfma_bf(ah_, b_[3 * (kc_) + 0], acc_);
#define FF_MMA2(b0_, b1_, ch_)                                                                                   \
    _Pragma("unroll") for (int kc = 0; kc < 2; ++kc) {                                                           \
        const unsigned char* ap = sB + c * PS + ((ch_) * 32 + kc * 16 + hh * 8) * 2;                             \
        const u32x4 ah = *reinterpret_cast<const u32x4*>(ap), am = *reinterpret_cast<const u32x4*>(ap + PLANE),  \
                    al = *reinterpret_cast<const u32x4*>(ap + 2 * PLANE);                                        \
        FF_SIX(acc0, ah, am, al, b0_, kc)                                                                        \
        FF_SIX(acc1, ah, am, al, b1_, kc)                                                                        \
    }
    F_STAMP(0)
    FF_LOAD2(p0, p1, 0)
    {
        // [x | att] rows -> planes in LDS. idx & 127 < 64 <=> even wave: the source is wave-uniform.
        const bool second = (wave & 1) != 0;
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + i * NT, row = idx >> 7, c4 = idx & 63;
            v[i] = gbuf_load4(second ? rAtt : rX, (unsigned)((m0 + row) * 256 + c4 * 4) * 4u, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + i * NT, row = idx >> 7, c4 = idx & 127;
            put4(sB, row * PS + c4 * 8, v[i]);
        }
    }
    __syncthreads();
    F_STAMP(1)

    // ---- h = [x | att] . W0^T : wave owns hidden columns [64 wave, 64 wave + 64)
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll 1
    for (int ch = 0; ch < 16; ch += 2) {
        FF_LOAD2(q0, q1, ch + 1)
        FF_MMA2(p0, p1, ch)
        if (ch + 2 < 16) FF_LOAD2(p0, p1, ch + 2)
        FF_MMA2(q0, q1, ch + 1)
    }
#undef FF_LOAD2
#undef FF_MMA2
    F_STAMP(2)

    // first chunk of W3 and the residual rows in flight behind the normalisation
    const unsigned vb3 = (unsigned)wave * TILE_BYTES + lane * 16u;
    u32x4 s0[6], s1[6];
#define FF_LOAD1(b_, ch_) \
    _Pragma("unroll") for (int g = 0; g < 6; ++g) b_[g] = __builtin_amdgcn_raw_buffer_load_b128(rW3, vb3, (unsigned)((ch_) * 6 + g) * 1024u, 0);
    FF_LOAD1(s0, 0)
    const int ocol = wave * 32 + c;
    const unsigned vres = (unsigned)((m0 + 4 * hh) * 256 + ocol) * 4u;
    float resid[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
        resid[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rX, vres, (unsigned)((r & 3) + 8 * (r >> 2)) * 1024u, 0));

    __syncthreads();   // every wave is done reading the input planes: the hidden rows go over them, in fp32
    {
        const int col0 = wave * 64 + c;
        const float bv0 = a.b0[col0], bv1 = a.b0[col0 + 32];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float* dst = sA + acc_row(r, hh) * LD + col0;
            float h0 = acc0[r] + bv0, h1 = acc1[r] + bv1;
            if constexpr (ACT == 1) { h0 = fmaxf(h0, 0.f); h1 = fmaxf(h1, 0.f); }
            dst[0] = h0;
            dst[32] = h1;
        }
    }
    __syncthreads();
    F_STAMP(3)
    {
        // 16 threads per row, thread `part` holds columns part * 4 + 64 i + {0..3}: LayerNorm(512) + GELU (ACT 0), then the cut into planes
        const int row = tid >> 4, part = tid & 15;
        const float* rp = sA + row * LD + part * 4;
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4*>(rp + 64 * i);
        if constexpr (ACT == 0) {
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            const float mean = sum * (1.f / 512.f);
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
                ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
            const float rstd = rsqrtf(ss * (1.f / 512.f) + 1e-5f);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 gg = *reinterpret_cast<const float4*>(a.ln_g + part * 4 + 64 * i);
                const float4 bb = *reinterpret_cast<const float4*>(a.ln_b + part * 4 + 64 * i);
                float4 y;
                y.x = v[i].x * rstd * gg.x + bb.x; y.y = v[i].y * rstd * gg.y + bb.y;
                y.z = v[i].z * rstd * gg.z + bb.z; y.w = v[i].w * rstd * gg.w + bb.w;
                y.x = 0.5f * y.x * (1.f + erff(y.x * 0.70710678118654752440f));
                y.y = 0.5f * y.y * (1.f + erff(y.y * 0.70710678118654752440f));
                y.z = 0.5f * y.z * (1.f + erff(y.z * 0.70710678118654752440f));
                y.w = 0.5f * y.w * (1.f + erff(y.w * 0.70710678118654752440f));
                v[i] = y;
            }
        }
        F_STAMP(4)
        __syncthreads();   // every fp32 row is in registers: the planes go over them
#pragma unroll
        for (int i = 0; i < 8; ++i) put4(sB, row * PS + (part * 4 + 64 * i) * 2, v[i]);
    }
    __syncthreads();
    F_STAMP(5)

    // ---- x += h . W3^T + b3 : wave owns output columns [32 wave, 32 wave + 32)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
#define FF_MMA1(b_, ch_)                                                                                         \
    _Pragma("unroll") for (int kc = 0; kc < 2; ++kc) {                                                           \
        const unsigned char* ap = sB + c * PS + ((ch_) * 32 + kc * 16 + hh * 8) * 2;                             \
        const u32x4 ah = *reinterpret_cast<const u32x4*>(ap), am = *reinterpret_cast<const u32x4*>(ap + PLANE),  \
                    al = *reinterpret_cast<const u32x4*>(ap + 2 * PLANE);                                        \
        FF_SIX(acc0, ah, am, al, b_, kc)                                                                         \
    }
#pragma unroll 1
    for (int ch = 0; ch < 16; ch += 2) {
        FF_LOAD1(s1, ch + 1)
        FF_MMA1(s0, ch)
        if (ch + 2 < 16) FF_LOAD1(s0, ch + 2)
        FF_MMA1(s1, ch + 1)
    }
#undef FF_LOAD1
#undef FF_MMA1
#undef FF_SIX
    F_STAMP(6)
    const float bv = a.b3[ocol];
#pragma unroll
    for (int r = 0; r < 16; ++r)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(resid[r] + (acc0[r] + bv)), rX, vres, (unsigned)((r & 3) + 8 * (r >> 2)) * 1024u, 0);
    F_STAMP(7)
}

// The same arithmetic with TWO blocks per CU (round 6). In-kernel stamps of the kernel above (profiles/r06_ffn_timeline.txt) showed its two products at
// the matrix pipe's rate and half of a block's 73 k cycles outside them - staging, the normalisation (erf: 35 vector instructions per element), barriers
// with HBM latency behind them - with nothing to run meanwhile: 100 KB of planes leave room for one block per CU. Here the planes cover HALF of the
// contraction at a time - [x], then over the same bytes [att]; hidden columns 0..255, then 256..511, whose values wait in registers (16 per thread) - so
// that a block needs the 66 KB of its fp32 hidden rows and no more, and the weight steps are one MFMA k chunk (12 registers per tile and buffer) so
// that it fits 128 registers: two blocks per CU, one's phases under the other's products. Every sum is formed in the order of the kernel above: the two
// are bit-identical (tests/test_gpu_kernels.py). With one block per CU (a launch of <= #CUs blocks) this form is the slower one (weight steps of 12
// MFMAs hide less latency): the launcher picks by the number of blocks.
template <int ACT>
__global__ __launch_bounds__(ff::NT, 4) void ffn_fused_split_kernel(FfnArgs a) {
    using namespace ff;
    extern __shared__ __attribute__((aligned(16))) unsigned char sB[];
    float* const sA = reinterpret_cast<float*>(sB);
    const int z = blockIdx.y, pair = z >> 1;
    if (a.active && a.active[pair * a.pstride] == 0) return;
    const int M = a.m_ptr ? a.m_ptr[pair * a.pstride + (z & 1)] : a.m_max;
    const int m0 = blockIdx.x * BM;
    if (m0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // in an SGPR: everything derived from it is wave-uniform
    const int c = lane & 31, hh = lane >> 5;

    const __amdgpu_buffer_rsrc_t rX = gmake_rsrc(a.x + (long)z * a.x_bstride, (unsigned)M * 1024u);
    const __amdgpu_buffer_rsrc_t rAtt = gmake_rsrc(a.att + (long)z * a.att_bstride, (unsigned)M * 1024u);
    const __amdgpu_buffer_rsrc_t rW0 = gmake_rsrc(a.w0p, 512u * 512u * 6u);
    const __amdgpu_buffer_rsrc_t rW3 = gmake_rsrc(a.w3p, 256u * 512u * 6u);

    // a weight step = one MFMA k chunk (16 of the 512 k) x three planes (x two column tiles in the first product): 16-byte loads, 1 KiB per wave each,
    // double-buffered one step (12 / 6 MFMAs per wave) ahead
    const unsigned vb0 = (unsigned)(2 * wave) * TILE_BYTES + lane * 16u;
    u32x4 p0[3], p1[3], q0[3], q1[3];
#define FF_LOAD2(b0_, b1_, kc_)                                                                                  \
    _Pragma("unroll") for (int g = 0; g < 3; ++g) {                                                              \
        b0_[g] = __builtin_amdgcn_raw_buffer_load_b128(rW0, vb0, (unsigned)((kc_) * 3 + g) * 1024u, 0);              \
        b1_[g] = __builtin_amdgcn_raw_buffer_load_b128(rW0, vb0 + TILE_BYTES, (unsigned)((kc_) * 3 + g) * 1024u, 0); \
    }
    // the six products of one 16-deep k chunk, small ones first: A planes (h, m, l) from LDS, B planes b[0, 1, 2]
#define FF_SIX(acc_, ah_, am_, al_, b_)         \
    acc_ = mfma_bf(ah_, b_[2], acc_);           \
    acc_ = mfma_bf(al_, b_[0], acc_);           \
    acc_ = mfma_bf(am_, b_[1], acc_);           \
    acc_ = mfma_bf(ah_, b_[1], acc_);           \
    acc_ = mfma_bf(am_, b_[0], acc_);           \
    acc_ = mfma_bf(ah_, b_[0], acc_);
    // k chunk `kc_` of 32 lies in the half-K planes at k = 16 (kc_ & 15)
#define FF_MMA2(b0_, b1_, kc_)                                                                                   \
    {                                                                                                            \
        const unsigned char* ap = sB + c * PS2 + (((kc_) & 15) * 16 + hh * 8) * 2;                                \
        const u32x4 ah = *reinterpret_cast<const u32x4*>(ap), am = *reinterpret_cast<const u32x4*>(ap + PLANE2),  \
                    al = *reinterpret_cast<const u32x4*>(ap + 2 * PLANE2);                                        \
        FF_SIX(acc0, ah, am, al, b0_)                                                                            \
        FF_SIX(acc1, ah, am, al, b1_)                                                                            \
    }
    F_STAMP(0)
    FF_LOAD2(p0, p1, 0)
    float4 v[8];
    // 32 rows x 256 floats of x, then of att: thread -> (row = idx >> 6, float4 idx & 63), 4 float4 per thread and half
#define FF_FETCH(rsrc_)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                      \
        const int idx = tid + i * NT;                                                                    \
        v[i] = gbuf_load4(rsrc_, (unsigned)((m0 + (idx >> 6)) * 256 + (idx & 63) * 4) * 4u, 0);          \
    }
#define FF_PUT_IN()                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                      \
        const int idx = tid + i * NT;                                                                    \
        put4<PLANE2>(sB, (idx >> 6) * PS2 + (idx & 63) * 8, v[i]);                                                \
    }
    FF_FETCH(rX)
    FF_PUT_IN()
    FF_FETCH(rAtt)            // in flight behind the first half of the product
    __syncthreads();
    F_STAMP(1)

    // ---- h = [x | att] . W0^T : wave owns hidden columns [64 wave, 64 wave + 64); first the 256 k of x, then - planes over the same bytes - of att
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll 1
    for (int kc = 0; kc < 16; kc += 2) {
        FF_LOAD2(q0, q1, kc + 1)
        FF_MMA2(p0, p1, kc)
        FF_LOAD2(p0, p1, kc + 2)
        FF_MMA2(q0, q1, kc + 1)
    }
    F_STAMP(2)
    __syncthreads();          // every wave is done with the planes of x
    FF_PUT_IN()
    __syncthreads();
    F_STAMP(3)
#pragma unroll 1
    for (int kc = 16; kc < 32; kc += 2) {
        FF_LOAD2(q0, q1, kc + 1)
        FF_MMA2(p0, p1, kc)
        if (kc + 2 < 32) FF_LOAD2(p0, p1, kc + 2)
        FF_MMA2(q0, q1, kc + 1)
    }
#undef FF_LOAD2
#undef FF_MMA2
#undef FF_FETCH
#undef FF_PUT_IN
    F_STAMP(4)

    // first step of W3 in flight behind the normalisation
    const unsigned vb3 = (unsigned)wave * TILE_BYTES + lane * 16u;
    u32x4 s0[3], s1[3];
#define FF_LOAD1(b_, kc_) \
    _Pragma("unroll") for (int g = 0; g < 3; ++g) b_[g] = __builtin_amdgcn_raw_buffer_load_b128(rW3, vb3, (unsigned)((kc_) * 3 + g) * 1024u, 0);
    FF_LOAD1(s0, 0)

    __syncthreads();   // every wave is done reading the input planes: the hidden rows go over them, in fp32
    {
        const int col0 = wave * 64 + c;
        const float bv0 = a.b0[col0], bv1 = a.b0[col0 + 32];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float* dst = sA + acc_row(r, hh) * LD + col0;
            float h0 = acc0[r] + bv0, h1 = acc1[r] + bv1;
            if constexpr (ACT == 1) { h0 = fmaxf(h0, 0.f); h1 = fmaxf(h1, 0.f); }
            dst[0] = h0;
            dst[32] = h1;
        }
    }
    __syncthreads();
    F_STAMP(5)
    // 16 threads per row, thread `part` holds columns part * 4 + 64 i + {0..3}: LayerNorm(512) + GELU (ACT 0), then the cut into planes.
    // The body below is the full-K kernel's, statement for statement, ON PURPOSE: as one shared device function the compiler contracts its multiplies and adds
    // differently (tried in round 6: both kernels still agreed with each other, but not with the previous build - tools/compare_builds.py)
    const int row = tid >> 4, part = tid & 15;
    {
        const float* rp = sA + row * LD + part * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4*>(rp + 64 * i);
        if constexpr (ACT == 0) {
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            const float mean = sum * (1.f / 512.f);
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
                ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
            const float rstd = rsqrtf(ss * (1.f / 512.f) + 1e-5f);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 gg = *reinterpret_cast<const float4*>(a.ln_g + part * 4 + 64 * i);
                const float4 bb = *reinterpret_cast<const float4*>(a.ln_b + part * 4 + 64 * i);
                float4 y;
                y.x = v[i].x * rstd * gg.x + bb.x; y.y = v[i].y * rstd * gg.y + bb.y;
                y.z = v[i].z * rstd * gg.z + bb.z; y.w = v[i].w * rstd * gg.w + bb.w;
                y.x = 0.5f * y.x * (1.f + erff(y.x * 0.70710678118654752440f));
                y.y = 0.5f * y.y * (1.f + erff(y.y * 0.70710678118654752440f));
                y.z = 0.5f * y.z * (1.f + erff(y.z * 0.70710678118654752440f));
                y.w = 0.5f * y.w * (1.f + erff(y.w * 0.70710678118654752440f));
                v[i] = y;
            }
        }
    }
    F_STAMP(6)
    __syncthreads();   // every fp32 row is in registers: the planes of hidden columns 0..255 go over them, 256..511 wait in registers
#pragma unroll
    for (int i = 0; i < 4; ++i) put4<PLANE2>(sB, row * PS2 + (part * 4 + 64 * i) * 2, v[i]);
    __syncthreads();
    F_STAMP(7)

    // ---- x += h . W3^T + b3 : wave owns output columns [32 wave, 32 wave + 32)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
#define FF_MMA1(b_, kc_)                                                                                         \
    {                                                                                                            \
        const unsigned char* ap = sB + c * PS2 + (((kc_) & 15) * 16 + hh * 8) * 2;                                \
        const u32x4 ah = *reinterpret_cast<const u32x4*>(ap), am = *reinterpret_cast<const u32x4*>(ap + PLANE2),  \
                    al = *reinterpret_cast<const u32x4*>(ap + 2 * PLANE2);                                        \
        FF_SIX(acc0, ah, am, al, b_)                                                                             \
    }
#pragma unroll 1
    for (int kc = 0; kc < 16; kc += 2) {
        FF_LOAD1(s1, kc + 1)
        FF_MMA1(s0, kc)
        FF_LOAD1(s0, kc + 2)
        FF_MMA1(s1, kc + 1)
    }
    F_STAMP(8)
    __syncthreads();          // every wave is done with the planes of the first 256 hidden columns
#pragma unroll
    for (int i = 0; i < 4; ++i) put4<PLANE2>(sB, row * PS2 + (part * 4 + 64 * i) * 2, v[4 + i]);
    // the residual rows in flight behind the second half
    const int ocol = wave * 32 + c;
    const unsigned vres = (unsigned)((m0 + 4 * hh) * 256 + ocol) * 4u;
    float resid[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
        resid[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rX, vres, (unsigned)((r & 3) + 8 * (r >> 2)) * 1024u, 0));
    __syncthreads();
    F_STAMP(9)
#pragma unroll 1
    for (int kc = 16; kc < 32; kc += 2) {
        FF_LOAD1(s1, kc + 1)
        FF_MMA1(s0, kc)
        if (kc + 2 < 32) FF_LOAD1(s0, kc + 2)
        FF_MMA1(s1, kc + 1)
    }
#undef FF_LOAD1
#undef FF_MMA1
#undef FF_SIX
    F_STAMP(10)
    const float bv = a.b3[ocol];
#pragma unroll
    for (int r = 0; r < 16; ++r)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(resid[r] + (acc0[r] + bv)), rX, vres, (unsigned)((r & 3) + 8 * (r >> 2)) * 1024u, 0);
    F_STAMP(11)
}

// Which form: two blocks per CU (half-K planes) when the launch has more blocks than the chip has CUs, one block per CU (full-K planes, weight steps of
// 24 MFMAs) otherwise - one pair of 4096 keypoints is 256 blocks. IM_FFN_SPLIT=0 | 1 forces one (read per call: the tests run both forms in one process
// and compare them bit for bit).
hipError_t launch_ffn_fused(const FfnArgs& a, hipStream_t s) {
    static size_t lds_optin[IM_MAX_DEVICES] = {0};   // per device: a process may hold contexts on several GPUs
    static size_t lds_optin_relu[IM_MAX_DEVICES] = {0};
    static size_t lds_optin2[IM_MAX_DEVICES] = {0};
    static size_t lds_optin2_relu[IM_MAX_DEVICES] = {0};
    static int n_cu[IM_MAX_DEVICES] = {0};
    if (a.m_max <= 0 || a.batch <= 0) return hipSuccess;
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    if (dev < 0 || dev >= IM_MAX_DEVICES) return hipErrorInvalidDevice;
    if (!n_cu[dev]) {
        if (hipError_t e = hipDeviceGetAttribute(&n_cu[dev], hipDeviceAttributeMultiprocessorCount, dev); e != hipSuccess) return e;
    }
    const dim3 grid((a.m_max + ff::BM - 1) / ff::BM, a.batch), block(ff::NT);
    const char* const split_env = getenv("IM_FFN_SPLIT");
    const bool split = split_env ? split_env[0] == '1' : (long)grid.x * grid.y > n_cu[dev];
    if (split) {
        if (a.act == 1) {
            if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(&ffn_fused_split_kernel<1>), ff::LDS_BYTES2, lds_optin2_relu); e != hipSuccess) return e;
            hipLaunchKernelGGL(ffn_fused_split_kernel<1>, grid, block, ff::LDS_BYTES2, s, a);
        } else {
            if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(&ffn_fused_split_kernel<0>), ff::LDS_BYTES2, lds_optin2); e != hipSuccess) return e;
            hipLaunchKernelGGL(ffn_fused_split_kernel<0>, grid, block, ff::LDS_BYTES2, s, a);
        }
        return hipGetLastError();
    }
    if (a.act == 1) {
        if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(&ffn_fused_kernel<1>), ff::LDS_BYTES, lds_optin_relu); e != hipSuccess) return e;
        hipLaunchKernelGGL(ffn_fused_kernel<1>, grid, block, ff::LDS_BYTES, s, a);
    } else {
        if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(&ffn_fused_kernel<0>), ff::LDS_BYTES, lds_optin); e != hipSuccess) return e;
        hipLaunchKernelGGL(ffn_fused_kernel<0>, grid, block, ff::LDS_BYTES, s, a);
    }
    return hipGetLastError();
}

}  // namespace im
