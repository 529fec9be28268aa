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
// L2/HBM once per slab instead of nine times. 73 KB LDS per block => two blocks per CU, so one block's
// staging overlaps the other's MFMAs. Wave w owns tile rows 2w, 2w+1 (x 32 columns) x 64 channels = 2 x 2
// MFMA tiles; the 2x2 pool partners are then (same lane, adjacent registers) x (the wave's two row tiles),
// so pooling is register-local.
#include "common.h"
#include "kernels.h"

namespace im {

static constexpr int TH = 8, TW = 32, CC = 16, CS = 20;  // tile, channel slab, LDS pixel stride (floats)
static constexpr int PH = TH + 2, PW = TW + 2;
static constexpr int CONV_LDS_FLOATS = PH * PW * CS + 9 * 64 * CS;

template <bool POOL>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sIn = smem;                  // [PH][PW][CS]
    float* sW = smem + PH * PW * CS;    // [9][64][CS]

    const int nslices = a.Cout / 64;
    const int b = blockIdx.z / nslices, co0 = (blockIdx.z % nslices) * 64;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const float* in = a.in + (long)b * a.H * a.W * a.Cin;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nslab = a.Cin / CC;
    for (int slab = 0; slab < nslab; ++slab) {
        __syncthreads();
        // input patch: PH*PW pixels x 4 float4
        for (int idx = tid; idx < PH * PW * 4; idx += 256) {
            const int pix = idx >> 2, c4 = idx & 3;
            const int py = pix / PW, px = pix - py * PW;
            const int gy = y0 + py - 1, gx = x0 + px - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                v = *reinterpret_cast<const float4*>(in + ((long)gy * a.W + gx) * a.Cin + slab * CC + c4 * 4);
            *reinterpret_cast<float4*>(sIn + pix * CS + c4 * 4) = v;
        }
        // weight slab: [9][64 of Cout][16]
        const float* wsrc = a.w + (long)slab * 9 * a.Cout * CC;
        for (int idx = tid; idx < 9 * 64 * 4; idx += 256) {
            const int row = idx >> 2, c4 = idx & 3;          // row = tap * 64 + co
            const int tap = row >> 6, co = row & 63;
            float4 v = *reinterpret_cast<const float4*>(wsrc + ((long)tap * a.Cout + co0 + co) * CC + c4 * 4);
            *reinterpret_cast<float4*>(sW + row * CS + c4 * 4) = v;
        }
        __syncthreads();

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

hipError_t launch_conv3x3(const ConvArgs& a, hipStream_t s) {
    if (a.Cin % CC != 0 || a.Cout % 64 != 0) return hipErrorInvalidValue;
    dim3 grid((a.W + TW - 1) / TW, (a.H + TH - 1) / TH, a.B * (a.Cout / 64)), block(256);
    const size_t lds = CONV_LDS_FLOATS * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mfma_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mfma_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (a.pool) hipLaunchKernelGGL(conv3x3_mfma_kernel<true>, grid, block, lds, s, a);
    else hipLaunchKernelGGL(conv3x3_mfma_kernel<false>, grid, block, lds, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// conv1a (Cin = 1): u8 gray -> x / 255 (true fp32 division == the reference's float64-divide-then-round for
// all 256 values, `matchers.py:1220`) -> 3x3 stencil x 64 channels, bias, ReLU -> NHWC.
// Thread = (pixel, 4 channels): 16 threads write one pixel's 256 contiguous bytes.
__global__ __launch_bounds__(256) void conv1a_kernel(const uint8_t* __restrict__ img, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ out,
                                                      int H, int W) {
    const int b = blockIdx.y;
    const int cg = threadIdx.x & 15;
    float4 wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const float4*>(w + t * 64 + cg * 4);
    const float4 bv = *reinterpret_cast<const float4*>(bias + cg * 4);
    const uint8_t* im = img + (long)b * H * W;
    const long npix = (long)H * W;
    for (long p0 = (long)blockIdx.x * 64; p0 < npix; p0 += (long)gridDim.x * 64)
    for (long pix = p0 + (threadIdx.x >> 4); pix < min(p0 + 64, npix); pix += 16) {
        const int y = (int)(pix / W), x = (int)(pix - (long)y * W);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int yy = y + dy - 1, xx = x + dx - 1;
                float p = 0.f;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) p = (float)im[(long)yy * W + xx] / 255.0f;
                const float4 wv = wt[dy * 3 + dx];
                acc.x = fmaf(p, wv.x, acc.x); acc.y = fmaf(p, wv.y, acc.y);
                acc.z = fmaf(p, wv.z, acc.z); acc.w = fmaf(p, wv.w, acc.w);
            }
        acc.x = fmaxf(acc.x + bv.x, 0.f); acc.y = fmaxf(acc.y + bv.y, 0.f);
        acc.z = fmaxf(acc.z + bv.z, 0.f); acc.w = fmaxf(acc.w + bv.w, 0.f);
        *reinterpret_cast<float4*>(out + ((long)b * npix + pix) * 64 + cg * 4) = acc;
    }
}

hipError_t launch_conv1a(const uint8_t* img, const float* w, const float* bias, float* out, int B, int H, int W,
                         hipStream_t s) {
    const long npix = (long)H * W;
    int gx = (int)((npix + 63) / 64);
    if (gx > 65535 * 4) gx = 65535 * 4;
    hipLaunchKernelGGL(conv1a_kernel, dim3(gx, B), dim3(256), 0, s, img, w, bias, out, H, W);
    return hipGetLastError();
}

}  // namespace im
