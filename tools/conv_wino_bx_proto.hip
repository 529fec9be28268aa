// Prototype (round 5, NOT part of the library): the Winograd F(2x2, 3x3) convolution of SuperPoint's 64 -> 64 layers with its sixteen
// element-wise products on the bf16 matrix cores at fp32 accuracy (six bf16 products per fp32 product, as csrc/attention_bx.hip), in the wave
// layout DESIGN.md section 9 proposes for round 6: eight waves per block, a wave owns TWO positions of one V row x 64 output channels.
// It answers one question before the product kernel (csrc/conv_wino.hip: slab pipeline, fused first layer, pooling, ragged edges) is rebuilt
// around that layout: what does the plain 64 -> 64 layer cost this way? Deliberately simple - the whole 64-channel halo patch of a region is
// loaded once, no pipelining across regions, the inverse transform through one LDS exchange - so the number is a floor for the layout's
// worth, not a tuned kernel.
//   hipcc -O3 --offload-arch=gfx950 tools/conv_wino_bx_proto.hip -o /tmp/conv_bx && /tmp/conv_bx
// prints the maximum error against a float64 direct convolution on a small image and the time of conv2a's shape (2 x 540 x 960 x 64 -> 64).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

static constexpr int C = 64;                      // input = output channels
#ifndef NG
#define NG 1                                         // groups of 32 tiles per wave: 1 = 8 x 16 output pixels per block, 2 = 16 x 16 (every U fragment serves twice the tiles)
#endif
static constexpr int TH = 8 * NG, TW = 16;        // output pixels per region: 4 NG x 8 Winograd tiles
static constexpr int PH = TH + 2, PW = TW + 2;    // halo patch
static constexpr int PIX = 68;                    // floats per patch pixel (64 + 4: adjacent tiles 8 banks apart)
static constexpr int PROW = 1284;                 // floats per patch row (18 x 68 = 1224, padded to 4 mod 64)
static constexpr int PATCH_BYTES = PH * PROW * 4; // 51,360
static constexpr int MS = 33;                     // floats per (position, tile) row of the exchange image: one 32-channel half at a time
static constexpr int M_BYTES = 16 * 32 * NG * MS * 4;  // 67,584 at NG = 1: two blocks per CU
static constexpr int LDS_BYTES = M_BYTES > PATCH_BYTES ? M_BYTES : PATCH_BYTES;

// Timing-only ablations (wrong results by construction), -DABL_NO_MFMA / -DABL_NO_CUT / -DABL_NO_READ / -DABL_NO_EPI: what each part of a block costs
__device__ __forceinline__ f32x16 mfma_bf(u32x4 a, u32x4 b, f32x16 c) {
#ifdef ABL_NO_MFMA
    c[0] += __uint_as_float(a.x ^ b.x);      // keeps the operands alive: one vector instruction instead of the MFMA
    return c;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
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
struct P3 { u32x4 h, m, l; };
__device__ __forceinline__ P3 split8(const float* x) {
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split2(x[2 * i], x[2 * i + 1], h[i], m[i], l[i]);
    return P3{u32x4{h[0], h[1], h[2], h[3]}, u32x4{m[0], m[1], m[2], m[3]}, u32x4{l[0], l[1], l[2], l[3]}};
}

// in / out: NHWC fp32 [B][H][W][64]; up: U planes packed on the host [position 16][chunk 4][cout tile 2][plane 3][lane 64][8] bf16
__global__ __launch_bounds__(512, NG == 1 ? 4 : 2) void conv_wino_bx(const float* __restrict__ in, const uint16_t* __restrict__ up, const float* __restrict__ bias,
                                                         float* __restrict__ out, int H, int W, int relu) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int rx = (W + TW - 1) / TW;
    const int x0 = (blockIdx.x % rx) * TW, y0 = (blockIdx.x / rx) * TH, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const float* img = in + (long)b * H * W * C;

    // ---- the 10 x 18 x 64 halo patch, zero outside the image
    {
        constexpr int NL = (PH * PW * 16 + 511) / 512;
        float4 v[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = tid + 512 * i, pix = idx >> 4, f4 = idx & 15, py = pix / PW, px = pix - py * PW;
            const int y = y0 - 1 + py, x = x0 - 1 + px;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#ifndef ABL_NO_PATCH
            if (idx < PH * PW * 16 && y >= 0 && y < H && x >= 0 && x < W) v[i] = *reinterpret_cast<const float4*>(img + ((long)y * W + x) * C + f4 * 4);
#endif
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = tid + 512 * i, pix = idx >> 4, f4 = idx & 15, py = pix / PW, px = pix - py * PW;
            if (idx < PH * PW * 16) *reinterpret_cast<float4*>(lds + py * PROW + px * PIX + f4 * 4) = v[i];
        }
    }
    __syncthreads();

    // ---- wave (ph, pp): V row ph, positions 2 pp and 2 pp + 1; lane (c, hh): tile c = (ty, tx), channels 8 hh .. 8 hh + 7 of the chunk
    const int ph = wave >> 1, pp = wave & 1;
    const int ty = c >> 3, tx = c & 7;
    const int ra = ph == 0 ? 0 : (ph == 2 ? 2 : 1), rb = ph == 0 ? 2 : (ph == 1 ? 2 : (ph == 2 ? 1 : 3));
    const float sgn = ph == 1 ? 1.f : -1.f;
    const float* pa = lds + (2 * ty + ra) * PROW + (2 * tx) * PIX + 8 * hh;
    const float* pb = lds + (2 * ty + rb) * PROW + (2 * tx) * PIX + 8 * hh;
    f32x16 acc[NG][2][2];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][q][ct][r] = 0.f;
    const u32x4* upl = reinterpret_cast<const u32x4*>(up) + lane;     // + ((((p * 4 + chunk) * 2 + ct) * 3 + plane) * 64) 16-byte pieces
    const int p0 = 4 * ph + 2 * pp;
    u32x4 ub[2][2][2][3];                                             // [buffer][position][cout tile][plane]
    auto load_u = [&](int buf, int chunk) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
#ifdef ABL_NO_U
                for (int pl = 0; pl < 3; ++pl) ub[buf][q][ct][pl] = u32x4{(unsigned)(lane + chunk), 2u + pl, 3u + q, 4u + ct};
#elif defined(ABL_ONE_U)   // timing-only ablation (wrong results): every chunk reads the SAME 12 KB of U - what is the L2 stream of the U planes worth?
                for (int pl = 0; pl < 3; ++pl) ub[buf][q][ct][pl] = upl[((((p0 + q) * 4 + 0 * chunk) * 2 + ct) * 3 + pl) * 64 * (1 + 0 * buf)];
#else
                for (int pl = 0; pl < 3; ++pl) ub[buf][q][ct][pl] = upl[((((p0 + q) * 4 + chunk) * 2 + ct) * 3 + pl) * 64];
#endif
    };
    load_u(0, 0);
#pragma unroll
    for (int chunk = 0; chunk < 4; ++chunk) {
        if (chunk < 3) load_u((chunk + 1) & 1, chunk + 1);
#pragma unroll
        for (int g = 0; g < NG; ++g) {       // tile group g: tiles 32 g + c = patch rows + 8 g
        float t[4][8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#ifdef ABL_NO_READ
            const float fz = (float)(chunk + j + g);
            const float4 a0 = make_float4(fz, sgn, fz, sgn), a1 = a0, b0 = make_float4(sgn, fz, fz, fz), b1 = b0;
#else
            const float4 a0 = *reinterpret_cast<const float4*>(pa + 8 * g * PROW + j * PIX + 16 * chunk), a1 = *reinterpret_cast<const float4*>(pa + 8 * g * PROW + j * PIX + 16 * chunk + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(pb + 8 * g * PROW + j * PIX + 16 * chunk), b1 = *reinterpret_cast<const float4*>(pb + 8 * g * PROW + j * PIX + 16 * chunk + 4);
#endif
            t[j][0] = a0.x + sgn * b0.x; t[j][1] = a0.y + sgn * b0.y; t[j][2] = a0.z + sgn * b0.z; t[j][3] = a0.w + sgn * b0.w;
            t[j][4] = a1.x + sgn * b1.x; t[j][5] = a1.y + sgn * b1.y; t[j][6] = a1.z + sgn * b1.z; t[j][7] = a1.w + sgn * b1.w;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (pp == 0) v[i] = q == 0 ? t[0][i] - t[2][i] : t[1][i] + t[2][i];
                else v[i] = q == 0 ? t[2][i] - t[1][i] : t[1][i] - t[3][i];
            }
#ifdef ABL_NO_CUT
            const P3 a = P3{u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])},
                            u32x4{__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])}, u32x4{1u, 2u, 3u, 4u}};
#else
            const P3 a = split8(v);
#endif
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const u32x4 bh = ub[chunk & 1][q][ct][0], bm = ub[chunk & 1][q][ct][1], bl = ub[chunk & 1][q][ct][2];
                f32x16& x = acc[g][q][ct];
                x = mfma_bf(a.h, bl, x);
                x = mfma_bf(a.l, bh, x);
                x = mfma_bf(a.m, bm, x);
                x = mfma_bf(a.h, bm, x);
                x = mfma_bf(a.m, bh, x);
                x = mfma_bf(a.h, bh, x);
            }
        }
        }
    }
    // ---- Y = A^T M A per (tile, cout), A^T = [1 1 1 0; 0 1 -1 -1], one 32-channel half at a time through the exchange image M[position][tile][32]
    float* img_out = out + (long)b * H * W * C;
#ifdef ABL_NO_EPI
    {
        float sum = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sum += acc[g][q][ct][r];
        if (sum == 12345.678f) img_out[tid] = sum;
        return;
    }
#endif
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        __syncthreads();       // the patch (ct = 0) / the first half (ct = 1) is dead
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tile = 32 * g + (r & 3) + 8 * (r >> 2) + 4 * hh;
                lds[((p0 + q) * 32 * NG + tile) * MS + c] = acc[g][q][ct][r];
            }
        __syncthreads();
#pragma unroll
        for (int i2 = 0; i2 < 2 * NG; ++i2) {
            const int item = tid + 512 * i2, tile = item >> 5, co = item & 31;
            float m[16];
#pragma unroll
            for (int p = 0; p < 16; ++p) m[p] = lds[(p * 32 * NG + tile) * MS + co];
            float s0[4], s1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { s0[j] = m[j] + m[4 + j] + m[8 + j]; s1[j] = m[4 + j] - m[8 + j] - m[12 + j]; }
            const float bv = bias[32 * ct + co];
            float y[2][2] = {{s0[0] + s0[1] + s0[2] + bv, s0[1] - s0[2] - s0[3] + bv}, {s1[0] + s1[1] + s1[2] + bv, s1[1] - s1[2] - s1[3] + bv}};
            const int oy = y0 + 2 * (tile >> 3), ox = x0 + 2 * (tile & 7);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (oy + i < H && ox + j < W) img_out[((long)(oy + i) * W + ox + j) * C + 32 * ct + co] = relu ? fmaxf(y[i][j], 0.f) : y[i][j];
        }
    }
}

// ---- version 3 (-DV3): the same layout with the block's parts overlapped: the halo patch in four 16-channel chunks through a double-buffered LDS image, the
// next chunk's global loads (and U planes) in flight under this chunk's products; the exchange with 8-byte accesses in a [position][cout][tile] image
static constexpr int P3PIX = 20;                       // floats per pixel of a patch chunk (16 + 4)
static constexpr int P3ROW = 364;                      // floats per patch row (18 x 20 = 360, + 4)
static constexpr int P3BUF = 10 * P3ROW;               // floats per buffer
static constexpr int M3S = 34;                         // floats per (position, cout) row of the exchange image: 32 tiles + 2 (8-byte reads of 32 couts conflict-free)
static constexpr int M3_BYTES = 16 * 32 * M3S * 4;     // 69,632 for one 32-cout half
static constexpr int LDS3_BYTES = M3_BYTES > 2 * P3BUF * 4 ? M3_BYTES : 2 * P3BUF * 4;

__global__ __launch_bounds__(512, 2) void conv_wino_bx3(const float* __restrict__ in, const uint16_t* __restrict__ up, const float* __restrict__ bias,
                                                          float* __restrict__ out, int H, int W, int relu) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int rx = (W + 15) / 16;
    const int x0 = (blockIdx.x % rx) * 16, y0 = (blockIdx.x / rx) * 8, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const float* img = in + (long)b * H * W * C;

    // patch chunk loads: 180 pixels x 4 float4 = 720 pieces, two per thread (the second one for 208 threads)
    int pofs[2], gofs[2]; bool pin[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 512 * i, pix = idx >> 2, f4 = idx & 3, py = pix / 18, px = pix - py * 18;
        const int y = y0 - 1 + py, x = x0 - 1 + px;
        pin[i] = idx < 720 && y >= 0 && y < H && x >= 0 && x < W;
        pofs[i] = idx < 720 ? py * P3ROW + px * P3PIX + f4 * 4 : -1;
        gofs[i] = pin[i] ? (y * W + x) * C + f4 * 4 : 0;
    }
    float4 pv[2];
    auto patch_load = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < 2; ++i) pv[i] = pin[i] ? *reinterpret_cast<const float4*>(img + gofs[i] + 16 * chunk) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto patch_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) if (pofs[i] >= 0) *reinterpret_cast<float4*>(lds + buf * P3BUF + pofs[i]) = pv[i];
    };

    const int ph = wave >> 1, pp = wave & 1;
    const int ty = c >> 3, tx = c & 7;
    const int ra = ph == 0 ? 0 : (ph == 2 ? 2 : 1), rb = ph == 0 ? 2 : (ph == 1 ? 2 : (ph == 2 ? 1 : 3));
    const float sgn = ph == 1 ? 1.f : -1.f;
    const int oa = (2 * ty + ra) * P3ROW + (2 * tx) * P3PIX + 8 * hh, ob = (2 * ty + rb) * P3ROW + (2 * tx) * P3PIX + 8 * hh;
    f32x16 acc[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][ct][r] = 0.f;
    const u32x4* upl = reinterpret_cast<const u32x4*>(up) + lane;
    const int p0 = 4 * ph + 2 * pp;
    u32x4 ub[2][2][2][3];
    auto load_u = [&](int buf, int chunk) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) ub[buf][q][ct][pl] = upl[((((p0 + q) * 4 + chunk) * 2 + ct) * 3 + pl) * 64];
    };
    patch_load(0);
    load_u(0, 0);
    patch_store(0);
    __syncthreads();
#pragma unroll
    for (int chunk = 0; chunk < 4; ++chunk) {
        if (chunk < 3) { patch_load(chunk + 1); load_u((chunk + 1) & 1, chunk + 1); }
        const float* pa = lds + (chunk & 1) * P3BUF + oa;
        const float* pb = lds + (chunk & 1) * P3BUF + ob;
        float t[4][8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 a0 = *reinterpret_cast<const float4*>(pa + j * P3PIX), a1 = *reinterpret_cast<const float4*>(pa + j * P3PIX + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(pb + j * P3PIX), b1 = *reinterpret_cast<const float4*>(pb + j * P3PIX + 4);
            t[j][0] = a0.x + sgn * b0.x; t[j][1] = a0.y + sgn * b0.y; t[j][2] = a0.z + sgn * b0.z; t[j][3] = a0.w + sgn * b0.w;
            t[j][4] = a1.x + sgn * b1.x; t[j][5] = a1.y + sgn * b1.y; t[j][6] = a1.z + sgn * b1.z; t[j][7] = a1.w + sgn * b1.w;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (pp == 0) v[i] = q == 0 ? t[0][i] - t[2][i] : t[1][i] + t[2][i];
                else v[i] = q == 0 ? t[2][i] - t[1][i] : t[1][i] - t[3][i];
            }
            const P3 a = split8(v);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const u32x4 bh = ub[chunk & 1][q][ct][0], bm = ub[chunk & 1][q][ct][1], bl = ub[chunk & 1][q][ct][2];
                f32x16& x = acc[q][ct];
                x = mfma_bf(a.h, bl, x);
                x = mfma_bf(a.l, bh, x);
                x = mfma_bf(a.m, bm, x);
                x = mfma_bf(a.h, bm, x);
                x = mfma_bf(a.m, bh, x);
                x = mfma_bf(a.h, bh, x);
            }
        }
        if (chunk < 3) {
            patch_store((chunk + 1) & 1);      // the other buffer: last read in chunk - 1, behind the barrier of that step
            __syncthreads();
        }
    }
    // ---- inverse transform, one 32-cout half at a time through M[position][cout][tile] (8-byte accesses)
    float* img_out = out + (long)b * H * W * C;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) {       // registers 4 g .. 4 g + 3 = tiles 8 g + 4 hh + (0..3)
                float* dst = lds + ((p0 + q) * 32 + c) * M3S + 8 * g + 4 * hh;
                *reinterpret_cast<float2*>(dst) = make_float2(acc[q][ct][4 * g], acc[q][ct][4 * g + 1]);
                *reinterpret_cast<float2*>(dst + 2) = make_float2(acc[q][ct][4 * g + 2], acc[q][ct][4 * g + 3]);
            }
        __syncthreads();
        {
            const int co = tid & 31, tp = tid >> 5;          // two tiles 2 tp, 2 tp + 1 of cout co
            float2 m[16];
#pragma unroll
            for (int p = 0; p < 16; ++p) m[p] = *reinterpret_cast<const float2*>(lds + (p * 32 + co) * M3S + 2 * tp);
            const float bv = bias[32 * ct + co];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float mm[16];
#pragma unroll
                for (int p = 0; p < 16; ++p) mm[p] = e ? m[p].y : m[p].x;
                float s0[4], s1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { s0[j] = mm[j] + mm[4 + j] + mm[8 + j]; s1[j] = mm[4 + j] - mm[8 + j] - mm[12 + j]; }
                float y[2][2] = {{s0[0] + s0[1] + s0[2] + bv, s0[1] - s0[2] - s0[3] + bv}, {s1[0] + s1[1] + s1[2] + bv, s1[1] - s1[2] - s1[3] + bv}};
                const int tile = 2 * tp + e, oy = y0 + 2 * (tile >> 3), ox = x0 + 2 * (tile & 7);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (oy + i < H && ox + j < W) img_out[((long)(oy + i) * W + ox + j) * C + 32 * ct + co] = relu ? fmaxf(y[i][j], 0.f) : y[i][j];
            }
        }
    }
}

static uint16_t host_bf16(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float host_bf16_f(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int main() {
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> w((size_t)C * C * 9), bias(C);
    for (auto& v : w) v = nd(rng) / std::sqrt(9.f * C);
    for (auto& v : bias) v = 0.1f * nd(rng);
    // U = G g G^T (double, rounded to fp32 as the library does), then the three planes in fragment order
    const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    std::vector<uint16_t> up((size_t)16 * 4 * 2 * 3 * 64 * 8);
    for (int co = 0; co < C; ++co)
        for (int ci = 0; ci < C; ++ci) {
            double g[3][3], tmp[4][3], U[4][4];
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) g[a][b] = w[((size_t)co * C + ci) * 9 + a * 3 + b];
            for (int i = 0; i < 4; ++i) for (int b = 0; b < 3; ++b) { tmp[i][b] = 0; for (int a = 0; a < 3; ++a) tmp[i][b] += G[i][a] * g[a][b]; }
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { U[i][j] = 0; for (int b = 0; b < 3; ++b) U[i][j] += tmp[i][b] * G[j][b]; }
            for (int p = 0; p < 16; ++p) {
                const float x = (float)U[p >> 2][p & 3];
                const uint16_t h = host_bf16(x);
                const float r1 = x - host_bf16_f(h);
                const uint16_t m = host_bf16(r1);
                const uint16_t l = host_bf16(r1 - host_bf16_f(m));
                const int chunk = ci >> 4, hh = (ci >> 3) & 1, j = ci & 7, ct = co >> 5, c = co & 31, lane = hh * 32 + c;
                const size_t base = ((((size_t)p * 4 + chunk) * 2 + ct) * 3) * 512 + (size_t)lane * 8 + j;
                up[base] = h; up[base + 512] = m; up[base + 1024] = l;
            }
        }
    uint16_t* d_up; float *d_bias, *d_in, *d_out;
    hipMalloc(&d_up, up.size() * 2); hipMemcpy(d_up, up.data(), up.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&d_bias, C * 4); hipMemcpy(d_bias, bias.data(), C * 4, hipMemcpyHostToDevice);
#ifdef V3
#define KERNEL conv_wino_bx3
#define KLDS LDS3_BYTES
#else
#define KERNEL conv_wino_bx
#define KLDS LDS_BYTES
#endif
    hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, KLDS);

    {   // ---- accuracy on a small ragged image against a float64 direct convolution
        const int B = 1, H = 21, W = 37;
        std::vector<float> x((size_t)B * H * W * C), y(x.size());
        for (auto& v : x) v = nd(rng);
        hipMalloc(&d_in, x.size() * 4); hipMalloc(&d_out, x.size() * 4);
        hipMemcpy(d_in, x.data(), x.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(KERNEL, dim3(((W + TW - 1) / TW) * ((H + TH - 1) / TH), B), dim3(512), KLDS, 0, d_in, d_up, d_bias, d_out, H, W, 0);
        hipMemcpy(y.data(), d_out, y.size() * 4, hipMemcpyDeviceToHost);
        double mx = 0, mxref = 0;
        for (int yy = 0; yy < H; ++yy)
            for (int xx = 0; xx < W; ++xx)
                for (int co = 0; co < C; ++co) {
                    double s = bias[co];
                    for (int a = 0; a < 3; ++a)
                        for (int bb = 0; bb < 3; ++bb) {
                            const int iy = yy + a - 1, ix = xx + bb - 1;
                            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                            for (int ci = 0; ci < C; ++ci) s += (double)x[((size_t)iy * W + ix) * C + ci] * w[((size_t)co * C + ci) * 9 + a * 3 + bb];
                        }
                    mx = std::fmax(mx, std::fabs(s - y[((size_t)yy * W + xx) * C + co]));
                    mxref = std::fmax(mxref, std::fabs(s));
                }
        printf("accuracy, 1 x 21 x 37 x 64 -> 64 against a float64 direct convolution: max abs error %.3e (max |y| %.2f)\n", mx, mxref);
        hipFree(d_in); hipFree(d_out);
    }
    {   // ---- time at conv2a's shape
        const int B = 2, H = 540, W = 960;
        std::vector<float> x((size_t)B * H * W * C);
        for (auto& v : x) v = nd(rng);
        hipMalloc(&d_in, x.size() * 4); hipMalloc(&d_out, x.size() * 4);
        hipMemcpy(d_in, x.data(), x.size() * 4, hipMemcpyHostToDevice);
        const dim3 grid(((W + TW - 1) / TW) * ((H + TH - 1) / TH), B);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(KERNEL, grid, dim3(512), KLDS, 0, d_in, d_up, d_bias, d_out, H, W, 1);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) {
                const double gf = 2.0 * 9 * C * C * H * W * B / 1e9;
                printf("time, %d tile group(s) per wave, 2 x 540 x 960 x 64 -> 64 (conv2a's shape; the library's f32-input MFMA kernel: 0.31 ms): %.4f ms = %.1f TFLOP/s direct-form\n", NG, ms / 20, gf / (ms / 20));
            }
        }
    }
    return 0;
}
