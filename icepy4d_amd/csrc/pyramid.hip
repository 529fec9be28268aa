// Gaussian pyramid steps on 8-bit images, on the device: what `Quality` resizing and tile preselection call through OpenCV in
// the reference (`cv2.pyrDown` / `cv2.pyrUp`, `matchers.py:529-530, 599-609`). OpenCV is an un-vendored dependency; the
// kernels restate its documented 8-bit algorithm (parity with a particular OpenCV build is unpinned):
//   pyrDown: separable [1 4 6 4 1] / 16 on BORDER_REFLECT_101-extended input, every second row / column kept,
//            integer accumulation, (sum + 128) >> 8; output ((H + 1) / 2) x ((W + 1) / 2)
//   pyrUp  : zero insertion and 4 x the same kernel: even samples (prev + 6 cur + next) / 8, odd samples (cur + next) / 2
//            per axis, integer accumulation, (sum + 32) >> 6; output 2H x 2W
// Pure byte traffic (HBM-bound): one thread per output byte, channels interleaved, consecutive lanes on consecutive bytes of
// an output row so that stores are dense and the strided tap reads of a wave fall into the same few cache lines.
#include "ctx.h"

namespace im {

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    i %= p;
    if (i < 0) i += p;
    return i >= n ? p - i : i;
}

__global__ __launch_bounds__(256) void pyr_down_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H, int W, int C, int OH, int OW) {
    const int xc = blockIdx.x * 256 + threadIdx.x;      // ox * C + channel
    const int oy = blockIdx.y, b = blockIdx.z;
    if (xc >= OW * C) return;
    const int ox = xc / C, ch = xc - ox * C;
    const uint8_t* img = in + (long)b * H * W * C;
    int xs[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) xs[d] = reflect101(2 * ox + d - 2, W) * C + ch;
    const int kw[5] = {1, 4, 6, 4, 1};
    int sum = 0;
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) {
        const uint8_t* row = img + (long)reflect101(2 * oy + dy - 2, H) * W * C;
        const int r = (int)row[xs[0]] + 4 * (int)row[xs[1]] + 6 * (int)row[xs[2]] + 4 * (int)row[xs[3]] + (int)row[xs[4]];
        sum += kw[dy] * r;
    }
    out[((long)b * OH + oy) * OW * C + xc] = (uint8_t)((sum + 128) >> 8);
}

__global__ __launch_bounds__(256) void pyr_up_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H, int W, int C) {
    const int OW = 2 * W, OH = 2 * H;
    const int xc = blockIdx.x * 256 + threadIdx.x;      // ox * C + channel
    const int oy = blockIdx.y, b = blockIdx.z;
    if (xc >= OW * C) return;
    const int ox = xc / C, ch = xc - ox * C;
    const uint8_t* img = in + (long)b * H * W * C;
    const int i = oy >> 1, j = ox >> 1;
    // per axis: even sample = prev + 6 cur + next, odd sample = 4 (cur + next). Borders as in OpenCV's pyrUp: the sample before
    // the first one is reflected (101), the one after the LAST one is the last one itself (its last column is
    // src[W-2] + 7 src[W-1] / 8 src[W-1]; rows go through borderInterpolate(2 sy, 2 H, REFLECT_101) / 2 = H - 1 for sy = H)
    int ry[3], wy[3], rx[3], wx[3];
    if (oy & 1) { ry[0] = i; wy[0] = 4; ry[1] = min(i + 1, H - 1); wy[1] = 4; ry[2] = i; wy[2] = 0; }
    else { ry[0] = reflect101(i - 1, H); wy[0] = 1; ry[1] = i; wy[1] = 6; ry[2] = min(i + 1, H - 1); wy[2] = 1; }
    if (ox & 1) { rx[0] = j; wx[0] = 4; rx[1] = min(j + 1, W - 1); wx[1] = 4; rx[2] = j; wx[2] = 0; }
    else { rx[0] = reflect101(j - 1, W); wx[0] = 1; rx[1] = j; wx[1] = 6; rx[2] = min(j + 1, W - 1); wx[2] = 1; }
    int sum = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const uint8_t* row = img + (long)ry[a] * W * C + ch;
        sum += wy[a] * (wx[0] * (int)row[rx[0] * C] + wx[1] * (int)row[rx[1] * C] + wx[2] * (int)row[rx[2] * C]);
    }
    out[((long)b * OH + oy) * OW * C + xc] = (uint8_t)min(max((sum + 32) >> 6, 0), 255);
}

}  // namespace im

using namespace im;

extern "C" {

int im_pyr_down(im_ctx* ctx, const uint8_t* d_in, uint8_t* d_out, int n_images, int h, int w, int channels, void* stream) {
    IM_CHECK_CTX(ctx);
    if (!d_in || !d_out || n_images < 1 || h < 1 || w < 1 || channels < 1 || channels > 4) return ctx->fail(-80, "im_pyr_down: bad arguments");
    const int oh = (h + 1) / 2, ow = (w + 1) / 2;
    if (oh > 65535 || n_images > 65535) return ctx->fail(-80, "im_pyr_down: image too large");
    hipLaunchKernelGGL(pyr_down_kernel, dim3((ow * channels + 255) / 256, oh, n_images), dim3(256), 0, (hipStream_t)stream, d_in, d_out, h, w,
                       channels, oh, ow);
    IM_HIP(ctx, hipGetLastError());
    return 0;
}

int im_pyr_up(im_ctx* ctx, const uint8_t* d_in, uint8_t* d_out, int n_images, int h, int w, int channels, void* stream) {
    IM_CHECK_CTX(ctx);
    if (!d_in || !d_out || n_images < 1 || h < 1 || w < 1 || channels < 1 || channels > 4) return ctx->fail(-81, "im_pyr_up: bad arguments");
    if (2 * h > 65535 || n_images > 65535) return ctx->fail(-81, "im_pyr_up: image too large");
    hipLaunchKernelGGL(pyr_up_kernel, dim3((2 * w * channels + 255) / 256, 2 * h, n_images), dim3(256), 0, (hipStream_t)stream, d_in, d_out, h, w,
                       channels);
    IM_HIP(ctx, hipGetLastError());
    return 0;
}

}  // extern "C"
