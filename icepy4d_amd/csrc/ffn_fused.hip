// LightGlue's feed-forward tail of a transformer block as ONE kernel (`lightglue/lightglue.py:144-149, 160-162, 212-216`):
//
//   x += W3 . gelu(layernorm(W0 . [x | att] + b0)) + b3          W0: 512 x 512 (out_proj / to_out folded in), W3: 256 x 512
//
// It replaces three launches (ffn.0 GEMM, LayerNorm + GELU, ffn.3 GEMM + residual) and the two HBM round trips of the 512-wide
// hidden rows (3 x 16.8 MB per launch at 2 x 4096 keypoints).
//
// A block owns 32 rows of one image and holds them in LDS for the whole kernel: first [x | att] (32 x 512 floats), then - in
// the same buffer - the hidden rows. 8 waves: in the first GEMM a wave owns 64 of the 512 hidden columns (two 32 x 32 MFMA
// tiles), in the second 32 of the 256 output columns (one tile). With a single row tile per block nothing of W is shared
// between the waves of a block, so W never goes through LDS: the weights are packed at load time in MFMA-fragment order
// ([column tile][k group of 8][lane][4 floats]: a wave's 16-byte-per-lane load is one contiguous KiB) and stream from L2
// straight into registers, double-buffered one 32-deep k chunk (32 MFMAs per wave) ahead. Between the four block barriers
// (inputs staged / first GEMM done / hidden rows normalised / - ) the waves run free of each other.
//
// LayerNorm is the two-pass form of lg_misc.hip's layernorm_gelu_kernel (mean, then centred sum of squares, eps 1e-5, erf GELU).
#include "common.h"
#include "kernels.h"
#include "sp_post.h"

#include <vector>

namespace im {

namespace ff {
constexpr int BM = 32;              // rows per block
constexpr int NT = 512;             // threads per block (8 waves)
constexpr int LD = 516;             // LDS row stride in floats: 516 = 4 (mod 32), 16-byte fragment reads are conflict free
constexpr int LDS_BYTES = BM * LD * 4;
constexpr unsigned TILE_BYTES = (512 / 8) * 64 * 16;   // one packed 32-column tile at K = 512
}  // namespace ff

// Fragment-order packing of a row-major W[N][K] (host side, once per weight set):
//   out[tile = n / 32][G = k / 8][lane = hh * 32 + c][i] = W[tile * 32 + c][8 G + 4 hh + i]
// MFMA step i of group G multiplies the k pair (8 G + i, 8 G + 4 + i); the A fragments are read with the same mapping.
std::vector<float> pack_frag_weights(const float* w, int n, int k) {
    std::vector<float> out((size_t)n * k);
    const int groups = k / 8;
    for (int t = 0; t < n / 32; ++t)
        for (int g = 0; g < groups; ++g)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 4; ++i)
                    out[(((size_t)t * groups + g) * 64 + lane) * 4 + i] = w[(size_t)(t * 32 + (lane & 31)) * k + 8 * g + 4 * (lane >> 5) + i];
    return out;
}

// ACT 0: LayerNorm(512) + GELU between the two products (LightGlue); ACT 1: ReLU (SuperGlue's MLP with its BatchNorm folded into
// W0 / b0, `SuperGlue/models/superglue.py:51-61, 104-116`)
template <int ACT>
__global__ __launch_bounds__(ff::NT, 4) void ffn_fused_kernel(FfnArgs a) {
    using namespace ff;
    extern __shared__ __attribute__((aligned(16))) float sA[];
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
    const __amdgpu_buffer_rsrc_t rW0 = gmake_rsrc(a.w0p, 512u * 512u * 4u);
    const __amdgpu_buffer_rsrc_t rW3 = gmake_rsrc(a.w3p, 256u * 512u * 4u);

    // ---- first weight chunk in flight while the input rows are staged
    const unsigned vb0 = (unsigned)(2 * wave) * TILE_BYTES + lane * 16u;
    float4 p0[4], p1[4], q0[4], q1[4];
#define FF_LOAD2(b0_, b1_, ch_)                                                     \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                 \
        b0_[g] = gbuf_load4(rW0, vb0, (unsigned)((ch_) * 4 + g) * 1024u);            \
        b1_[g] = gbuf_load4(rW0, vb0 + TILE_BYTES, (unsigned)((ch_) * 4 + g) * 1024u); \
    }
#define FF_MMA2(b0_, b1_, ch_)                                                                                  \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                             \
        const float4 fa = *reinterpret_cast<const float4*>(sA + c * LD + ((ch_) * 4 + g) * 8 + hh * 4);         \
        acc0 = mfma32(fa.x, b0_[g].x, acc0); acc1 = mfma32(fa.x, b1_[g].x, acc1);                               \
        acc0 = mfma32(fa.y, b0_[g].y, acc0); acc1 = mfma32(fa.y, b1_[g].y, acc1);                               \
        acc0 = mfma32(fa.z, b0_[g].z, acc0); acc1 = mfma32(fa.z, b1_[g].z, acc1);                               \
        acc0 = mfma32(fa.w, b0_[g].w, acc0); acc1 = mfma32(fa.w, b1_[g].w, acc1);                               \
    }
    FF_LOAD2(p0, p1, 0)
    {
        // [x | att] rows -> LDS. idx & 127 < 64 <=> even wave: the source is wave-uniform.
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
            *reinterpret_cast<float4*>(sA + row * LD + c4 * 4) = v[i];
        }
    }
    __syncthreads();

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

    // first chunk of W3 and the residual rows in flight behind the normalisation
    const unsigned vb3 = (unsigned)wave * TILE_BYTES + lane * 16u;
    float4 s0[4], s1[4];
#define FF_LOAD1(b_, ch_) \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) b_[g] = gbuf_load4(rW3, vb3, (unsigned)((ch_) * 4 + g) * 1024u);
    FF_LOAD1(s0, 0)
    const int ocol = wave * 32 + c;
    const unsigned vres = (unsigned)((m0 + 4 * hh) * 256 + ocol) * 4u;
    float resid[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
        resid[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rX, vres, (unsigned)((r & 3) + 8 * (r >> 2)) * 1024u, 0));

    __syncthreads();   // every wave is done reading the input rows
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
    if constexpr (ACT == 0) {
        // LayerNorm(512) + GELU in place: 16 threads per row, thread `part` holds columns part * 4 + 64 i + {0..3}
        const int row = tid >> 4, part = tid & 15;
        float* rp = sA + row * LD + part * 4;
        float4 v[8];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[i] = *reinterpret_cast<const float4*>(rp + 64 * i);
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
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
            *reinterpret_cast<float4*>(rp + 64 * i) = y;
        }
        __syncthreads();
    }

    // ---- x += h . W3^T + b3 : wave owns output columns [32 wave, 32 wave + 32)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
#define FF_MMA1(b_, ch_)                                                                                        \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                             \
        const float4 fa = *reinterpret_cast<const float4*>(sA + c * LD + ((ch_) * 4 + g) * 8 + hh * 4);         \
        acc0 = mfma32(fa.x, b_[g].x, acc0);                                                                     \
        acc0 = mfma32(fa.y, b_[g].y, acc0);                                                                     \
        acc0 = mfma32(fa.z, b_[g].z, acc0);                                                                     \
        acc0 = mfma32(fa.w, b_[g].w, acc0);                                                                     \
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
    const float bv = a.b3[ocol];
#pragma unroll
    for (int r = 0; r < 16; ++r)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(resid[r] + (acc0[r] + bv)), rX, vres, (unsigned)((r & 3) + 8 * (r >> 2)) * 1024u, 0);
}

hipError_t launch_ffn_fused(const FfnArgs& a, hipStream_t s) {
    static size_t lds_optin[IM_MAX_DEVICES] = {0};   // per device: a process may hold contexts on several GPUs
    static size_t lds_optin_relu[IM_MAX_DEVICES] = {0};
    if (a.m_max <= 0 || a.batch <= 0) return hipSuccess;
    const dim3 grid((a.m_max + ff::BM - 1) / ff::BM, a.batch), block(ff::NT);
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
