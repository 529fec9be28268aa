// SuperPoint VGG encoder convolutions (`lightglue/superpoint.py:155-165, 168, 203`; twin
// `SuperGlue/models/superpoint.py:154-164`): 3x3, stride 1, zero padding 1, bias, ReLU, optional fused
// 2x2/2 max-pool, as an implicit GEMM on the f32-input matrix cores.
//
//   M = pixels of an 8 x 32 spatial tile, N = 64 output channels, K = 9 taps x Cin
//
// Layout: activations NHWC fp32 (a pixel's channels are contiguous, so the halo patch is staged into LDS
// with plain 16-byte copies and MFMA A-fragments are 16-byte LDS reads); weights pre-packed on the host as
// [Cin/16][tap][Cout][16]. Per 16-channel slab the block stages the (8+2) x (32+2) input patch and the
// 9 x 64 x 16 weight slab, then runs 9 taps x 8 k-steps of MFMAs out of LDS: each input value is read from
// L2/HBM once per slab instead of nine times. The NEXT slab's patch and weights are fetched into registers
// while the current slab's MFMAs run (branch-free: out-of-image pixels are clamped loads turned into zeros by a
// select), and written to LDS after them. 73 KB LDS per block => two blocks per CU.
// Wave w owns tile rows 2w, 2w+1 (x 32 columns) x 64 channels = 2 x 2 MFMA tiles; the 2x2 pool partners are
// then (same lane, adjacent registers) x (the wave's two row tiles), so pooling is register-local.
//
// FUSE1A (first layer pair conv1a -> conv1b): the patch is not loaded but computed on the fly from the uint8 image
// (x / 255, 3x3 stencil with conv1a's 64 filters, bias, ReLU; `matchers.py:1220`, `superpoint.py:155`), so the
// full-resolution 64-channel conv1a activation (1 GB per 1080p pair) never exists in HBM.
#include "common.h"
#include "kernels.h"
#include "sp_post.h"

namespace im {

static constexpr int TH = 8, TW = 32, CC = 16, CS = 20;  // tile, channel slab, LDS pixel stride (floats)
static constexpr int PH = TH + 2, PW = TW + 2;
static constexpr int NPIX4 = PH * PW * 4;                 // float4 items of one patch slab (1360)
static constexpr int IH = TH + 4, IW = TW + 4;            // uint8 image tile needed by the fused conv1a (halo 2)
static constexpr int CONV_LDS_FLOATS = PH * PW * CS + 9 * 64 * CS;
static constexpr int FUSE_LDS_FLOATS = IH * IW + 9 * 64 + 64;  // image tile (as float), conv1a weights [9][64], bias

template <bool POOL, bool FUSE1A>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sIn = smem;                  // [PH][PW][CS]
    float* sW = smem + PH * PW * CS;    // [9][64][CS]
    float* sImg = smem + CONV_LDS_FLOATS;          // FUSE1A only: [IH][IW]
    float* sW1 = sImg + IH * IW;                   // [9][64]
    float* sB1 = sW1 + 9 * 64;                     // [64]

    // XCD-aware block -> tile map: linear id = tile_lo (3 bits) + 8 * (slice + nslices * tile_hi). The 64-channel
    // output slices of one spatial tile share id % 8, i.e. the XCD, and are adjacent in launch order, so the input
    // patch they all read is served by one L2.
    const int nslices = a.Cout / 64;
    const int tx = (a.W + TW - 1) / TW, ty = (a.H + TH - 1) / TH;
    const int ntile = tx * ty * a.B;
    const int bid = blockIdx.x;
    const int tile = (bid & 7) + 8 * ((bid >> 3) / nslices);
    const int co0 = (((bid >> 3) % nslices)) * 64;
    if (tile >= ntile) return;
    const int b = tile / (tx * ty);
    const int trem = tile - b * tx * ty;
    const int x0 = (trem % tx) * TW, y0 = (trem / tx) * TH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const float* in = a.in + (long)b * a.H * a.W * a.Cin;

    if constexpr (FUSE1A) {
        const uint8_t* img = a.img + (long)b * a.H * a.W * a.img_channels;
        for (int idx = tid; idx < IH * IW; idx += 256) {
            const int iy = idx / IW, ix = idx - iy * IW;
            const int gy = y0 + iy - 2, gx = x0 + ix - 2;
            float v = 0.f;  // conv1a's own zero padding
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = image_value(img, (long)gy * a.W + gx, a.img_channels, a.gray_mode);
            sImg[idx] = v;
        }
        for (int idx = tid; idx < 9 * 64 + 64; idx += 256) sW1[idx] = idx < 576 ? a.w1[idx] : a.b1[idx - 576];
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- per-thread staging items: 6 patch float4 (the 6th only for tid < 80) and 9 weight float4
    float4 p0 = {}, p1 = {}, p2 = {}, p3 = {}, p4 = {}, p5 = {}, w0, w1, w2, w3, w4, w5, w6, w7, w8;

    // patch item `idx` of slab `slab`: pixel idx >> 2 of the (PH x PW) patch, channels 4 * (idx & 3) .. + 3
    auto patch_item = [&](int idx, int slab) -> float4 {
        const int pix = min(idx, NPIX4 - 1) >> 2, c4 = idx & 3;
        const int py = pix / PW, px = pix - py * PW;
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        float4 v;
        if constexpr (FUSE1A) {
            const int ch = slab * CC + c4 * 4;
            v = *reinterpret_cast<const float4*>(sB1 + ch);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float pv = sImg[(py + dy) * IW + px + dx];
                    const float4 wv = *reinterpret_cast<const float4*>(sW1 + (dy * 3 + dx) * 64 + ch);
                    v.x = fmaf(pv, wv.x, v.x); v.y = fmaf(pv, wv.y, v.y); v.z = fmaf(pv, wv.z, v.z); v.w = fmaf(pv, wv.w, v.w);
                }
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        } else {
            const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gx, 0), a.W - 1);
            v = *reinterpret_cast<const float4*>(in + ((long)cy * a.W + cx) * a.Cin + slab * CC + c4 * 4);
        }
        if (!inside) v = make_float4(0.f, 0.f, 0.f, 0.f);
        return v;
    };
    auto weight_item = [&](int idx, int slab) -> float4 {  // row = tap * 64 + co of the slab's [9][64][16] block
        const int row = idx >> 2, c4 = idx & 3;
        return *reinterpret_cast<const float4*>(a.w + ((long)slab * 9 * a.Cout + (long)(row >> 6) * a.Cout + co0 + (row & 63)) * CC + c4 * 4);
    };
    // FUSE1A: the patch has no global-memory latency to hide (its inputs sit in LDS), so it is computed at commit
    // time straight into LDS instead of being carried in registers across the MFMA block.
#define IM_FETCH(slab)                                                                                    \
    if constexpr (!FUSE1A) {                                                                              \
        p0 = patch_item(tid, slab); p1 = patch_item(tid + 256, slab); p2 = patch_item(tid + 512, slab);    \
        p3 = patch_item(tid + 768, slab); p4 = patch_item(tid + 1024, slab); p5 = patch_item(tid + 1280, slab); \
    }                                                                                                     \
    w0 = weight_item(tid, slab); w1 = weight_item(tid + 256, slab); w2 = weight_item(tid + 512, slab);      \
    w3 = weight_item(tid + 768, slab); w4 = weight_item(tid + 1024, slab); w5 = weight_item(tid + 1280, slab); \
    w6 = weight_item(tid + 1536, slab); w7 = weight_item(tid + 1792, slab); w8 = weight_item(tid + 2048, slab);
#define IM_PUT_P(i, r) *reinterpret_cast<float4*>(sIn + ((tid + (i) * 256) >> 2) * CS + (tid & 3) * 4) = r
#define IM_PUT_W(i, r) *reinterpret_cast<float4*>(sW + ((tid + (i) * 256) >> 2) * CS + (tid & 3) * 4) = r
#define IM_COMMIT(slab)                                                                          \
    if constexpr (FUSE1A) {                                                                      \
        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_)                                         \
            if (tid + i_ * 256 < NPIX4) IM_PUT_P(i_, patch_item(tid + i_ * 256, slab));          \
    } else {                                                                                     \
        IM_PUT_P(0, p0); IM_PUT_P(1, p1); IM_PUT_P(2, p2); IM_PUT_P(3, p3); IM_PUT_P(4, p4);     \
        if (tid + 1280 < NPIX4) IM_PUT_P(5, p5);                                                 \
    }                                                                                            \
    IM_PUT_W(0, w0); IM_PUT_W(1, w1); IM_PUT_W(2, w2); IM_PUT_W(3, w3); IM_PUT_W(4, w4);         \
    IM_PUT_W(5, w5); IM_PUT_W(6, w6); IM_PUT_W(7, w7); IM_PUT_W(8, w8);

    if constexpr (FUSE1A) __syncthreads();  // image tile + conv1a weights visible
    const int nslab = a.Cin / CC;
    IM_FETCH(0)
    for (int slab = 0; slab < nslab; ++slab) {
        __syncthreads();  // previous slab fully consumed
        IM_COMMIT(slab)
        __syncthreads();
        if (slab + 1 < nslab) { IM_FETCH(slab + 1) }

#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float* ap = sIn + ((2 * wave + dy) * PW + c + dx) * CS + hh * 8;
                const float* bp = sW + ((dy * 3 + dx) * 64 + c) * CS + hh * 8;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float4 a0 = *reinterpret_cast<const float4*>(ap + t * 4);
                    const float4 a1 = *reinterpret_cast<const float4*>(ap + PW * CS + t * 4);
                    const float4 b0 = *reinterpret_cast<const float4*>(bp + t * 4);
                    const float4 b1 = *reinterpret_cast<const float4*>(bp + 32 * CS + t * 4);
#define IM_STEP(e)                                      \
    acc[0][0] = mfma32(a0.e, b0.e, acc[0][0]);          \
    acc[0][1] = mfma32(a0.e, b1.e, acc[0][1]);          \
    acc[1][0] = mfma32(a1.e, b0.e, acc[1][0]);          \
    acc[1][1] = mfma32(a1.e, b1.e, acc[1][1]);
                    IM_STEP(x) IM_STEP(y) IM_STEP(z) IM_STEP(w)
#undef IM_STEP
                }
            }
        }
    }
#undef IM_FETCH
#undef IM_PUT_P
#undef IM_PUT_W
#undef IM_COMMIT

    // ---- epilogue. acc[i][j][r]: pixel (y0 + 2*wave + i, x0 + acc_row(r, hh)), channel co0 + 32*j + c
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = co0 + 32 * j + c;
        const float bv = a.bias[co];
        if constexpr (POOL) {
            const int Ho = a.H >> 1, Wo = a.W >> 1;
            const int oy = (y0 >> 1) + wave;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float v = fmaxf(fmaxf(acc[0][j][r], acc[0][j][r + 1]), fmaxf(acc[1][j][r], acc[1][j][r + 1])) + bv;
                if (a.relu) v = fmaxf(v, 0.f);
                const int ox = (x0 >> 1) + (acc_row(r, hh) >> 1);
                if (oy < Ho && ox < Wo) a.out[(((long)b * Ho + oy) * Wo + ox) * a.Cout + co] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int oy = y0 + 2 * wave + i;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] + bv;
                    if (a.relu) v = fmaxf(v, 0.f);
                    const int ox = x0 + acc_row(r, hh);
                    if (oy < a.H && ox < a.W) a.out[(((long)b * a.H + oy) * a.W + ox) * a.Cout + co] = v;
                }
            }
        }
    }
}

template <bool POOL, bool FUSE>
static hipError_t launch_conv_variant(const ConvArgs& a, hipStream_t s) {
    const int ntile = ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH) * a.B;
    dim3 grid(((ntile + 7) / 8) * 8 * (a.Cout / 64)), block(256);
    const size_t lds = (CONV_LDS_FLOATS + (FUSE ? FUSE_LDS_FLOATS : 0)) * sizeof(float);
    static size_t lds_optin[IM_MAX_DEVICES] = {0};   // per device: a process may hold contexts on several GPUs
    if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(&conv3x3_mfma_kernel<POOL, FUSE>), lds, lds_optin); e != hipSuccess) return e;
    hipLaunchKernelGGL((conv3x3_mfma_kernel<POOL, FUSE>), grid, block, lds, s, a);
    return hipGetLastError();
}

hipError_t launch_conv3x3(const ConvArgs& a, hipStream_t s) {
    if (a.Cin % CC != 0 || a.Cout % 64 != 0) return hipErrorInvalidValue;
    if (a.img) {
        if (a.Cin != 64 || !a.w1 || !a.b1) return hipErrorInvalidValue;
        return a.pool ? launch_conv_variant<true, true>(a, s) : launch_conv_variant<false, true>(a, s);
    }
    return a.pool ? launch_conv_variant<true, false>(a, s) : launch_conv_variant<false, false>(a, s);
}

// ---------------------------------------------------------------------------------------------
// conv1a alone (Cin = 1): u8 gray -> x / 255 (true fp32 division == the reference's float64-divide-then-round for
// all 256 values, `matchers.py:1220`) -> 3x3 stencil x 64 channels, bias, ReLU -> NHWC. Only used by the stage
// tests; the forward pass fuses it into conv1b (FUSE1A above).
// Thread = (pixel, 4 channels): 16 threads write one pixel's 256 contiguous bytes.
__global__ __launch_bounds__(256) void conv1a_kernel(const uint8_t* __restrict__ img, int channels, int gray_mode, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ out,
                                                      int H, int W) {
    const int b = blockIdx.y;
    const int cg = threadIdx.x & 15;
    float4 wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const float4*>(w + t * 64 + cg * 4);
    const float4 bv = *reinterpret_cast<const float4*>(bias + cg * 4);
    const uint8_t* im = img + (long)b * H * W * channels;
    const long npix = (long)H * W;
    for (long p0 = (long)blockIdx.x * 64; p0 < npix; p0 += (long)gridDim.x * 64)
    for (long pix = p0 + (threadIdx.x >> 4); pix < min(p0 + 64, npix); pix += 16) {
        const int y = (int)(pix / W), x = (int)(pix - (long)y * W);
        float4 acc = bv;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int yy = y + dy - 1, xx = x + dx - 1;
                float p = 0.f;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) p = image_value(im, (long)yy * W + xx, channels, gray_mode);
                const float4 wv = wt[dy * 3 + dx];
                acc.x = fmaf(p, wv.x, acc.x); acc.y = fmaf(p, wv.y, acc.y);
                acc.z = fmaf(p, wv.z, acc.z); acc.w = fmaf(p, wv.w, acc.w);
            }
        acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f);
        acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
        *reinterpret_cast<float4*>(out + ((long)b * npix + pix) * 64 + cg * 4) = acc;
    }
}

hipError_t launch_conv1a(const uint8_t* img, int channels, int gray_mode, const float* w, const float* bias, float* out, int B,
                         int H, int W, hipStream_t s) {
    const long npix = (long)H * W;
    int gx = (int)((npix + 63) / 64);
    if (gx > 65535 * 4) gx = 65535 * 4;
    hipLaunchKernelGGL(conv1a_kernel, dim3(gx, B), dim3(256), 0, s, img, channels, gray_mode, w, bias, out, H, W);
    return hipGetLastError();
}

}  // namespace im
