// Winograd F(2x2, 3x3) form of the SuperPoint 3x3 convolutions (same layers as conv.hip: `lightglue/superpoint.py:155-168,
// 203`), still on the f32-input matrix cores: 16 element-wise products per 2x2 output tile instead of 36, i.e. 2.25x
// fewer MFMA FLOPs than the direct implicit GEMM for the same result in exact arithmetic (fp32 rounding differs in the
// last bits, like any change of summation order; checked against an fp64 reference in tests/test_gpu_kernels.py).
//
//   V = B^T d B   (4x4 input tile d, per channel)          B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
//   U = G g G^T   (3x3 filter g, transformed on the host)  G   = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
//   M[p] = sum_cin V[p] * U[p]   for the 16 positions p    -> 16 GEMMs  [tiles x Cin] x [Cin x Cout]
//   Y = A^T M A   (2x2 outputs)                            A^T = [1 1 1 0; 0 1 -1 -1]
//
// Block = 512 threads = 8 waves, output region 8 x 32 pixels = 64 Winograd tiles x 64 output channels; wave w owns
// positions 2w, 2w+1 for all tiles and channels (2 x 2 x 2 MFMA tiles of 32 x 32 = 128 accumulator registers, so two
// waves fit per SIMD). Per 8-channel slab: the (8+2) x (32+2) halo patch and the slab's U block [16][64][8] are staged
// through registers into LDS, every thread transforms one (tile, channel) 4x4 patch into the double-buffered V image,
// and the MFMAs of slab s run in the same barrier interval as the transform of slab s+1. The inverse transform, bias,
// ReLU and the optional 2x2 max-pool (exactly one Winograd tile) happen once at the end through an LDS image of M.
// FUSE1A: the patch is conv1a(img / 255) computed on the fly, as in conv.hip.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace im {

static constexpr int WTH = 8, WTW = 32;               // output region of a block
static constexpr int WPH = WTH + 2, WPW = WTW + 2;    // halo patch
static constexpr int WNT = (WTH / 2) * (WTW / 2);     // 64 tiles
static constexpr int WCC = 8;                         // channels per slab
static constexpr int W_SP = WPH * WPW * WCC;          // floats: patch
static constexpr int W_SV = 16 * WNT * WCC;           // floats: one V image
static constexpr int W_SU = 16 * 64 * WCC;            // floats: one U block
static constexpr int W_MAIN = W_SP + 2 * W_SV + 2 * W_SU;
static constexpr int W_SM = 16 * WNT * 32;            // floats: M image of one 32-channel half (epilogue, aliases the above)
static constexpr int W_IH = WTH + 4, W_IW = WTW + 4;
static constexpr int W_FUSE = W_IH * W_IW + 9 * 64 + 64;
static constexpr int W_LDS_FLOATS = (W_MAIN > W_SM ? W_MAIN : W_SM);

template <bool POOL, bool FUSE1A>
__global__ __launch_bounds__(512, 2) void conv3x3_wino_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sP = smem;
    float* sV = smem + W_SP;            // [2][16][64][8]
    float* sU = sV + 2 * W_SV;          // [2][16][64][8]
    float* sM = smem;                   // epilogue alias [16][64][32]
    float* sImg = smem + W_LDS_FLOATS;  // FUSE1A only
    float* sW1 = sImg + W_IH * W_IW;
    float* sB1 = sW1 + 9 * 64;

    const int nslices = a.Cout / 64;
    const int tx = (a.W + WTW - 1) / WTW, ty = (a.H + WTH - 1) / WTH;
    const int ntile = tx * ty * a.B;
    const int bid = blockIdx.x;
    const int rtile = (bid & 7) + 8 * ((bid >> 3) / nslices);   // XCD-aware: slices of one region share bid % 8
    const int co0 = ((bid >> 3) % nslices) * 64;
    if (rtile >= ntile) return;
    const int b = rtile / (tx * ty);
    const int trem = rtile - b * tx * ty;
    const int x0 = (trem % tx) * WTW, y0 = (trem / tx) * WTH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const float* in = a.in + (long)b * a.H * a.W * a.Cin;

    if constexpr (FUSE1A) {
        const uint8_t* img = a.img + (long)b * a.H * a.W;
        for (int idx = tid; idx < W_IH * W_IW; idx += 512) {
            const int iy = idx / W_IW, ix = idx - iy * W_IW;
            const int gy = y0 + iy - 2, gx = x0 + ix - 2;
            float v = 0.f;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = (float)img[(long)gy * a.W + gx] / 255.0f;
            sImg[idx] = v;
        }
        for (int idx = tid; idx < 9 * 64 + 64; idx += 512) sW1[idx] = idx < 576 ? a.w1[idx] : a.b1[idx - 576];
        __syncthreads();
    }

    f32x16 acc[2][2][2];  // [position of this wave][tile block][cout block]
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][i][j][r] = 0.f;

    // patch item idx of a slab: pixel idx >> 1 of the patch, channels 4 * (idx & 1) .. + 3   (680 items)
    auto patch_item = [&](int idx, int slab) -> float4 {
        const int pix = min(idx, WPH * WPW * 2 - 1) >> 1, c4 = idx & 1;
        const int py = pix / WPW, px = pix - py * WPW;
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        float4 v;
        if constexpr (FUSE1A) {
            const int ch = slab * WCC + c4 * 4;
            v = *reinterpret_cast<const float4*>(sB1 + ch);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float pv = sImg[(py + dy) * W_IW + px + dx];
                    const float4 wv = *reinterpret_cast<const float4*>(sW1 + (dy * 3 + dx) * 64 + ch);
                    v.x = fmaf(pv, wv.x, v.x); v.y = fmaf(pv, wv.y, v.y); v.z = fmaf(pv, wv.z, v.z); v.w = fmaf(pv, wv.w, v.w);
                }
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        } else {
            const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gx, 0), a.W - 1);
            v = *reinterpret_cast<const float4*>(in + ((long)cy * a.W + cx) * a.Cin + slab * WCC + c4 * 4);
        }
        if (!inside) v = make_float4(0.f, 0.f, 0.f, 0.f);
        return v;
    };
    // U item idx of a slab: row = idx >> 1 = pos * 64 + co of the slab's [16][Cout][8] block, channels 4 * (idx & 1) ..
    auto u_item = [&](int idx, int slab) -> float4 {
        const int row = idx >> 1, c4 = idx & 1;
        return *reinterpret_cast<const float4*>(a.w + (((long)slab * 16 + (row >> 6)) * a.Cout + co0 + (row & 63)) * WCC + c4 * 4);
    };
    float4 p0 = {}, p1 = {}, u0, u1, u2, u3;
#define IM_WFETCH(slab)                                                                      \
    if constexpr (!FUSE1A) { p0 = patch_item(tid, slab); p1 = patch_item(tid + 512, slab); } \
    u0 = u_item(tid, slab); u1 = u_item(tid + 512, slab); u2 = u_item(tid + 1024, slab); u3 = u_item(tid + 1536, slab);
#define IM_WPUT(base, i, r) *reinterpret_cast<float4*>((base) + (tid + (i) * 512) * 4) = r
#define IM_WCOMMIT(slab)                                                                     \
    {                                                                                        \
        float* ub = sU + ((slab) & 1) * W_SU;                                                \
        if constexpr (FUSE1A) {                                                              \
            IM_WPUT(sP, 0, patch_item(tid, slab));                                           \
            if (tid + 512 < WPH * WPW * 2) IM_WPUT(sP, 1, patch_item(tid + 512, slab));      \
        } else {                                                                             \
            IM_WPUT(sP, 0, p0);                                                              \
            if (tid + 512 < WPH * WPW * 2) IM_WPUT(sP, 1, p1);                               \
        }                                                                                    \
        IM_WPUT(ub, 0, u0); IM_WPUT(ub, 1, u1); IM_WPUT(ub, 2, u2); IM_WPUT(ub, 3, u3);      \
    }
    // input transform of one (tile, channel): thread t -> tile t >> 3, channel t & 7; V image [pos][tile][8]
    const int t_tile = tid >> 3, t_ch = tid & 7;
    const int t_ty = t_tile >> 4, t_tx = t_tile & 15;
#define IM_WTRANSFORM(slab)                                                                  \
    {                                                                                        \
        float* vb = sV + ((slab) & 1) * W_SV + t_tile * WCC + t_ch;                          \
        const float* pp = sP + ((2 * t_ty) * WPW + 2 * t_tx) * WCC + t_ch;                   \
        float d[4][4];                                                                       \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                     \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) d[i_][j_] = pp[(i_ * WPW + j_) * WCC]; \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                   \
            const float r0 = d[0][j_] - d[2][j_], r1 = d[1][j_] + d[2][j_];                  \
            const float r2 = d[2][j_] - d[1][j_], r3 = d[1][j_] - d[3][j_];                  \
            d[0][j_] = r0; d[1][j_] = r1; d[2][j_] = r2; d[3][j_] = r3;                      \
        }                                                                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                   \
            vb[(i_ * 4 + 0) * WNT * WCC] = d[i_][0] - d[i_][2];                              \
            vb[(i_ * 4 + 1) * WNT * WCC] = d[i_][1] + d[i_][2];                              \
            vb[(i_ * 4 + 2) * WNT * WCC] = d[i_][2] - d[i_][1];                              \
            vb[(i_ * 4 + 3) * WNT * WCC] = d[i_][1] - d[i_][3];                              \
        }                                                                                    \
    }
#define IM_WMMA(slab)                                                                        \
    {                                                                                        \
        _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) {                                   \
            const float* vq = sV + ((slab) & 1) * W_SV + ((2 * wave + q_) * WNT + c) * WCC + hh * 4; \
            const float* uq = sU + ((slab) & 1) * W_SU + ((2 * wave + q_) * 64 + c) * WCC + hh * 4;  \
            const float4 a0 = *reinterpret_cast<const float4*>(vq);                          \
            const float4 a1 = *reinterpret_cast<const float4*>(vq + 32 * WCC);               \
            const float4 b0 = *reinterpret_cast<const float4*>(uq);                          \
            const float4 b1 = *reinterpret_cast<const float4*>(uq + 32 * WCC);               \
            IM_WSTEP(q_, x) IM_WSTEP(q_, y) IM_WSTEP(q_, z) IM_WSTEP(q_, w)                  \
        }                                                                                    \
    }
#define IM_WSTEP(q_, e)                                         \
    acc[q_][0][0] = mfma32(a0.e, b0.e, acc[q_][0][0]);          \
    acc[q_][0][1] = mfma32(a0.e, b1.e, acc[q_][0][1]);          \
    acc[q_][1][0] = mfma32(a1.e, b0.e, acc[q_][1][0]);          \
    acc[q_][1][1] = mfma32(a1.e, b1.e, acc[q_][1][1]);

    const int nslab = a.Cin / WCC;
    IM_WFETCH(0)
    IM_WCOMMIT(0)
    if (nslab > 1) { IM_WFETCH(1) }
    __syncthreads();
    IM_WTRANSFORM(0)
    __syncthreads();
    for (int slab = 0; slab < nslab; ++slab) {
        if (slab + 1 < nslab) {
            IM_WCOMMIT(slab + 1)           // patch + U of slab s+1 (sP was last read by the transform of slab s)
            if (slab + 2 < nslab) { IM_WFETCH(slab + 2) }
        }
        __syncthreads();
        // The two waves that share a SIMD (w and w + 4) take the interval's two jobs in opposite order, so one wave's
        // transform VALU/LDS work runs under the other's MFMAs instead of both queueing on the same pipe.
        if (wave < 4) {
            if (slab + 1 < nslab) IM_WTRANSFORM(slab + 1)
            IM_WMMA(slab)
        } else {
            IM_WMMA(slab)
            if (slab + 1 < nslab) IM_WTRANSFORM(slab + 1)
        }
        __syncthreads();
    }
#undef IM_WFETCH
#undef IM_WPUT
#undef IM_WCOMMIT
#undef IM_WTRANSFORM
#undef IM_WMMA
#undef IM_WSTEP

    // ---- inverse transform + bias + ReLU (+ pool), one 32-channel half at a time through the M image [16][64][32]
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        if (cb) __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    sM[((2 * wave + q) * WNT + tb * 32 + acc_row(r, hh)) * 32 + c] = acc[q][tb][cb][r];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + it * 512;
            const int co_l = idx & 31, tile = idx >> 5;
            const int tyy = tile >> 4, txx = tile & 15;
            float m[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) m[i][j] = sM[((i * 4 + j) * WNT + tile) * 32 + co_l];
            float s0[4], s1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0[j] = (m[0][j] + m[1][j]) + m[2][j];
                s1[j] = (m[1][j] - m[2][j]) - m[3][j];
            }
            const int co = co0 + cb * 32 + co_l;
            const float bv = a.bias[co];
            float y00 = (s0[0] + s0[1]) + s0[2] + bv, y01 = (s0[1] - s0[2]) - s0[3] + bv;
            float y10 = (s1[0] + s1[1]) + s1[2] + bv, y11 = (s1[1] - s1[2]) - s1[3] + bv;
            if (a.relu) { y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f); }
            const int oy = y0 + 2 * tyy, ox = x0 + 2 * txx;
            if constexpr (POOL) {
                const int Ho = a.H >> 1, Wo = a.W >> 1;
                const int py = oy >> 1, px = ox >> 1;
                if (py < Ho && px < Wo)
                    a.out[(((long)b * Ho + py) * Wo + px) * a.Cout + co] = fmaxf(fmaxf(y00, y01), fmaxf(y10, y11));
            } else {
                float* o = a.out + (((long)b * a.H + oy) * a.W + ox) * a.Cout + co;
                if (oy < a.H && ox < a.W) o[0] = y00;
                if (oy < a.H && ox + 1 < a.W) o[a.Cout] = y01;
                if (oy + 1 < a.H && ox < a.W) o[(long)a.W * a.Cout] = y10;
                if (oy + 1 < a.H && ox + 1 < a.W) o[(long)a.W * a.Cout + a.Cout] = y11;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Variant B: 256 threads = 4 waves; wave (tb, cb) owns ALL 16 Winograd positions of tile block tb (32 tiles) x channel
// block cb (32 channels): 16 x 16 = 256 accumulator registers, one wave per SIMD. Same LDS images and slab pipeline as
// above, but the inverse transform A^T M A is register-local (a lane holds one output channel, a register one tile, and
// the 16 positions are 16 accumulators), so the epilogue needs no LDS round trip and no barrier. Measured equal to
// variant A (conv1b at 1080p: 2.0 ms both); ablations of this variant (IM_ABL_*): without MFMAs 1.21 ms, without the
// transform 1.88, without staging 1.81, without the epilogue 1.95, prologue + one slab + epilogue only 0.49 ms, i.e.
// the MFMA share is 0.87 ms and nothing overlaps it yet at one block per CU (selected with IM_WINO_VARIANT=B).
template <bool POOL, bool FUSE1A>
__global__ __launch_bounds__(256) void conv3x3_wino_kernel_b(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sP = smem;
    float* sV = smem + W_SP;
    float* sU = sV + 2 * W_SV;
    float* sImg = smem + W_MAIN;
    float* sW1 = sImg + W_IH * W_IW;
    float* sB1 = sW1 + 9 * 64;

    const int nslices = a.Cout / 64;
    const int tx = (a.W + WTW - 1) / WTW, ty = (a.H + WTH - 1) / WTH;
    const int ntile = tx * ty * a.B;
    const int bid = blockIdx.x;
    const int rtile = (bid & 7) + 8 * ((bid >> 3) / nslices);
    const int co0 = ((bid >> 3) % nslices) * 64;
    if (rtile >= ntile) return;
    const int b = rtile / (tx * ty);
    const int trem = rtile - b * tx * ty;
    const int x0 = (trem % tx) * WTW, y0 = (trem / tx) * WTH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const int tb = wave & 1, cb = wave >> 1;
    const float* in = a.in + (long)b * a.H * a.W * a.Cin;

    if constexpr (FUSE1A) {
        const uint8_t* img = a.img + (long)b * a.H * a.W;
        for (int idx = tid; idx < W_IH * W_IW; idx += 256) {
            const int iy = idx / W_IW, ix = idx - iy * W_IW;
            const int gy = y0 + iy - 2, gx = x0 + ix - 2;
            float v = 0.f;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = (float)img[(long)gy * a.W + gx] / 255.0f;
            sImg[idx] = v;
        }
        for (int idx = tid; idx < 9 * 64 + 64; idx += 256) sW1[idx] = idx < 576 ? a.w1[idx] : a.b1[idx - 576];
        __syncthreads();
    }

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    auto patch_item = [&](int idx, int slab) -> float4 {
        const int pix = min(idx, WPH * WPW * 2 - 1) >> 1, c4 = idx & 1;
        const int py = pix / WPW, px = pix - py * WPW;
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        float4 v;
        if constexpr (FUSE1A) {
            const int ch = slab * WCC + c4 * 4;
            v = *reinterpret_cast<const float4*>(sB1 + ch);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float pv = sImg[(py + dy) * W_IW + px + dx];
                    const float4 wv = *reinterpret_cast<const float4*>(sW1 + (dy * 3 + dx) * 64 + ch);
                    v.x = fmaf(pv, wv.x, v.x); v.y = fmaf(pv, wv.y, v.y); v.z = fmaf(pv, wv.z, v.z); v.w = fmaf(pv, wv.w, v.w);
                }
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        } else {
            const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gx, 0), a.W - 1);
            v = *reinterpret_cast<const float4*>(in + ((long)cy * a.W + cx) * a.Cin + slab * WCC + c4 * 4);
        }
        if (!inside) v = make_float4(0.f, 0.f, 0.f, 0.f);
        return v;
    };
    auto u_item = [&](int idx, int slab) -> float4 {
        const int row = idx >> 1, c4 = idx & 1;
        return *reinterpret_cast<const float4*>(a.w + (((long)slab * 16 + (row >> 6)) * a.Cout + co0 + (row & 63)) * WCC + c4 * 4);
    };
    float4 p0 = {}, p1 = {}, p2 = {}, u0, u1, u2, u3, u4, u5, u6, u7;
#define IM_BFETCH(slab)                                                                                          \
    if constexpr (!FUSE1A) { p0 = patch_item(tid, slab); p1 = patch_item(tid + 256, slab); p2 = patch_item(tid + 512, slab); } \
    u0 = u_item(tid, slab); u1 = u_item(tid + 256, slab); u2 = u_item(tid + 512, slab); u3 = u_item(tid + 768, slab);  \
    u4 = u_item(tid + 1024, slab); u5 = u_item(tid + 1280, slab); u6 = u_item(tid + 1536, slab); u7 = u_item(tid + 1792, slab);
#define IM_BPUT(base, i, r) *reinterpret_cast<float4*>((base) + (tid + (i) * 256) * 4) = r
#define IM_BCOMMIT(slab)                                                                     \
    {                                                                                        \
        float* ub = sU + ((slab) & 1) * W_SU;                                                \
        if constexpr (FUSE1A) {                                                              \
            IM_BPUT(sP, 0, patch_item(tid, slab)); IM_BPUT(sP, 1, patch_item(tid + 256, slab)); \
            if (tid + 512 < WPH * WPW * 2) IM_BPUT(sP, 2, patch_item(tid + 512, slab));      \
        } else {                                                                             \
            IM_BPUT(sP, 0, p0); IM_BPUT(sP, 1, p1);                                          \
            if (tid + 512 < WPH * WPW * 2) IM_BPUT(sP, 2, p2);                               \
        }                                                                                    \
        IM_BPUT(ub, 0, u0); IM_BPUT(ub, 1, u1); IM_BPUT(ub, 2, u2); IM_BPUT(ub, 3, u3);      \
        IM_BPUT(ub, 4, u4); IM_BPUT(ub, 5, u5); IM_BPUT(ub, 6, u6); IM_BPUT(ub, 7, u7);      \
    }
    // input transform: thread t handles (tile, channel) items t and t + 256
#define IM_BTRANSFORM1(slab, item)                                                           \
    {                                                                                        \
        const int tt_ = (item) >> 3, tc_ = (item) & 7;                                       \
        float* vb = sV + ((slab) & 1) * W_SV + tt_ * WCC + tc_;                              \
        const float* pp = sP + ((2 * (tt_ >> 4)) * WPW + 2 * (tt_ & 15)) * WCC + tc_;        \
        float d[4][4];                                                                       \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                     \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) d[i_][j_] = pp[(i_ * WPW + j_) * WCC]; \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                   \
            const float r0 = d[0][j_] - d[2][j_], r1 = d[1][j_] + d[2][j_];                  \
            const float r2 = d[2][j_] - d[1][j_], r3 = d[1][j_] - d[3][j_];                  \
            d[0][j_] = r0; d[1][j_] = r1; d[2][j_] = r2; d[3][j_] = r3;                      \
        }                                                                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                   \
            vb[(i_ * 4 + 0) * WNT * WCC] = d[i_][0] - d[i_][2];                              \
            vb[(i_ * 4 + 1) * WNT * WCC] = d[i_][1] + d[i_][2];                              \
            vb[(i_ * 4 + 2) * WNT * WCC] = d[i_][2] - d[i_][1];                              \
            vb[(i_ * 4 + 3) * WNT * WCC] = d[i_][1] - d[i_][3];                              \
        }                                                                                    \
    }
#define IM_BTRANSFORM(slab) IM_BTRANSFORM1(slab, tid) IM_BTRANSFORM1(slab, tid + 256)
#define IM_BMMA(slab)                                                                        \
    {                                                                                        \
        const float* vq = sV + ((slab) & 1) * W_SV + (tb * 32 + c) * WCC + hh * 4;           \
        const float* uq = sU + ((slab) & 1) * W_SU + (cb * 32 + c) * WCC + hh * 4;           \
        _Pragma("unroll") for (int p_ = 0; p_ < 16; ++p_) {                                  \
            const float4 av = *reinterpret_cast<const float4*>(vq + p_ * WNT * WCC);         \
            const float4 bv = *reinterpret_cast<const float4*>(uq + p_ * 64 * WCC);          \
            acc[p_] = mfma32(av.x, bv.x, acc[p_]); acc[p_] = mfma32(av.y, bv.y, acc[p_]);    \
            acc[p_] = mfma32(av.z, bv.z, acc[p_]); acc[p_] = mfma32(av.w, bv.w, acc[p_]);    \
        }                                                                                    \
    }

    const int nslab = a.Cin / WCC;
    IM_BFETCH(0)
    IM_BCOMMIT(0)
    if (nslab > 1) { IM_BFETCH(1) }
    __syncthreads();
    IM_BTRANSFORM(0)
    __syncthreads();
#ifdef IM_ABL_NO_LOOP
    for (int slab = 0; slab + 1 < 1; ++slab) {
#else
    for (int slab = 0; slab + 1 < nslab; ++slab) {
#endif
#ifndef IM_ABL_NO_STAGE
        IM_BCOMMIT(slab + 1)
        if (slab + 2 < nslab) { IM_BFETCH(slab + 2) }
#endif
        __syncthreads();
#ifndef IM_ABL_NO_MMA
        IM_BMMA(slab)
#endif
#ifndef IM_ABL_NO_TRANSFORM
        IM_BTRANSFORM(slab + 1)
#endif
        __syncthreads();
    }
    IM_BMMA(nslab - 1)
#undef IM_BFETCH
#undef IM_BPUT
#undef IM_BCOMMIT
#undef IM_BTRANSFORM1
#undef IM_BTRANSFORM
#undef IM_BMMA

#ifdef IM_ABL_NO_EPI
    { float keep = 0.f;
      _Pragma("unroll") for (int p_ = 0; p_ < 16; ++p_) keep += acc[p_][0];
      if (keep == 123.456f) a.out[0] = keep;
      return; }
#endif
    // ---- register-local inverse transform: lane = output channel, register r = tile tb*32 + acc_row(r, hh)
    const int co = co0 + cb * 32 + c;
    const float bv = a.bias[co];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int tile = tb * 32 + acc_row(r, hh);
        const int tyy = tile >> 4, txx = tile & 15;
        float s0[4], s1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s0[j] = (acc[0 + j][r] + acc[4 + j][r]) + acc[8 + j][r];
            s1[j] = (acc[4 + j][r] - acc[8 + j][r]) - acc[12 + j][r];
        }
        float y00 = (s0[0] + s0[1]) + s0[2] + bv, y01 = (s0[1] - s0[2]) - s0[3] + bv;
        float y10 = (s1[0] + s1[1]) + s1[2] + bv, y11 = (s1[1] - s1[2]) - s1[3] + bv;
        if (a.relu) { y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f); }
        const int oy = y0 + 2 * tyy, ox = x0 + 2 * txx;
        if constexpr (POOL) {
            const int Ho = a.H >> 1, Wo = a.W >> 1;
            const int py = oy >> 1, px = ox >> 1;
            if (py < Ho && px < Wo) a.out[(((long)b * Ho + py) * Wo + px) * a.Cout + co] = fmaxf(fmaxf(y00, y01), fmaxf(y10, y11));
        } else {
            float* o = a.out + (((long)b * a.H + oy) * a.W + ox) * a.Cout + co;
            if (oy < a.H && ox < a.W) o[0] = y00;
            if (oy < a.H && ox + 1 < a.W) o[a.Cout] = y01;
            if (oy + 1 < a.H && ox < a.W) o[(long)a.W * a.Cout] = y10;
            if (oy + 1 < a.H && ox + 1 < a.W) o[(long)a.W * a.Cout + a.Cout] = y11;
        }
    }
}


template <bool POOL, bool FUSE>
static hipError_t launch_wino_variant(const ConvArgs& a, hipStream_t s) {
    const int ntile = ((a.W + WTW - 1) / WTW) * ((a.H + WTH - 1) / WTH) * a.B;
    static const char variant = getenv("IM_WINO_VARIANT") ? getenv("IM_WINO_VARIANT")[0] : 'A';  // A (default) / B: tuning switch
    if (variant == 'A') {
        dim3 grid(((ntile + 7) / 8) * 8 * (a.Cout / 64)), block(512);
        const size_t lds = (W_LDS_FLOATS + (FUSE ? W_FUSE : 0)) * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel<POOL, FUSE>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            attr_set = true;
        }
        hipLaunchKernelGGL((conv3x3_wino_kernel<POOL, FUSE>), grid, block, lds, s, a);
        return hipGetLastError();
    }
    dim3 grid(((ntile + 7) / 8) * 8 * (a.Cout / 64)), block(256);
    const size_t lds = (W_MAIN + (FUSE ? W_FUSE : 0)) * sizeof(float);
    static bool attr_set_b = false;
    if (!attr_set_b) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel_b<POOL, FUSE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set_b = true;
    }
    hipLaunchKernelGGL((conv3x3_wino_kernel_b<POOL, FUSE>), grid, block, lds, s, a);
    return hipGetLastError();
}

// a.w must be the Winograd-packed weights [Cin/8][16][Cout][8] (pack_conv3x3_wino)
hipError_t launch_conv3x3_wino(const ConvArgs& a, hipStream_t s) {
    if (a.Cin % WCC != 0 || a.Cout % 64 != 0) return hipErrorInvalidValue;
    if (a.img) {
        if (a.Cin != 64 || !a.w1 || !a.b1) return hipErrorInvalidValue;
        return a.pool ? launch_wino_variant<true, true>(a, s) : launch_wino_variant<false, true>(a, s);
    }
    return a.pool ? launch_wino_variant<true, false>(a, s) : launch_wino_variant<false, false>(a, s);
}

}  // namespace im
