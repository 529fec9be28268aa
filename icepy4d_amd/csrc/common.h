// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of libicematch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define IM_WAVE 64
#define IM_MAX_DEVICES 16

// v_mfma_f32_32x32x2_f32: D[32x32] += A[32x2] * B[2x32], exact fp32 (k-ordered fmaf chain).
// Lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31].
// D layout: lane l holds column j = l & 31, rows (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), r = 0..15.
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// row index held by accumulator register r of a 32x32 MFMA tile for lane-half hh
__device__ __forceinline__ int acc_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}


// Raw buffer access: a descriptor sized to the live bytes (reads past it return zero, stores past it are dropped), lane
// offsets in a VGPR, the wave-uniform part of the address in an SGPR - no 64-bit address arithmetic on the VALU.
typedef unsigned int gu32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 gbuf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    const gu32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t gmake_rsrc(const void* base, unsigned bytes) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    void* p = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// Image value fed to conv1a for pixel `pix` of a uint8 image with 1 or 3 interleaved channels.
//   1 channel : x / 255 (`matchers.py:1212-1220`, `:263-274`; true fp32 division == float64 divide then round, for all 256 inputs)
//   3 channels, gray_mode 0 (LightGlue flavour): the reference scales the HWC image to float first and converts on the float
//     image: kornia.color.rgb_to_grayscale = w_r * r + w_g * g + w_b * b with fp32 weights (0.299, 0.587, 0.114), three
//     separate tensor multiplies and two adds, each rounded to fp32 (`lightglue/utils.py:35-36`): no fused multiply-add here
//   3 channels, gray_mode 1 (SuperGlue flavour): cv2.cvtColor(RGB2GRAY) on the uint8 image (`matchers.py:911-914`): OpenCV's
//     fixed-point form with 14 fractional bits, (4899 R + 9617 G + 1868 B + 8192) >> 14, then x / 255
//   4 "channels" = 4 bytes per pixel: a float32 gray image the host has already scaled (and resized: `lightglue/utils.py:30-33`)
// kornia and cv2 are un-vendored: both formulas restate their published code (parity unpinned at these two call sites).
__device__ __forceinline__ float image_value(const uint8_t* __restrict__ img, long pix, int channels, int gray_mode) {
    if (channels == 1) return (float)img[pix] / 255.0f;
    if (channels == 4) return reinterpret_cast<const float*>(img)[pix];   // float32 gray, already scaled (the `resize` path)
    const uint8_t* p = img + pix * 3;
    const unsigned r = p[0], g = p[1], b = p[2];
    if (gray_mode == 1) return (float)((r * 4899u + g * 9617u + b * 1868u + 8192u) >> 14) / 255.0f;
    const float rf = (float)r / 255.0f, gf = (float)g / 255.0f, bf = (float)b / 255.0f;
    return __fadd_rn(__fadd_rn(__fmul_rn(0.299f, rf), __fmul_rn(0.587f, gf)), __fmul_rn(0.114f, bf));
}

// order-preserving float -> uint32 map (larger float <=> larger uint)
__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// torch log_sigmoid: min(x, 0) - log1p(exp(-|x|))
__device__ __forceinline__ float log_sigmoid(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }
