// Experiment behind DESIGN.md section 8 ("beyond the fp32 roofline"): how accurate is an fp32 product computed on the
// BF16 matrix cores from a 3-way split of each operand (x = x0 + x1 + x2, 8 significand bits each, fp32 accumulation)?
//   C = A B^T with A [M][K], B [N][K] fp32, computed three ways and compared with an fp64 reference:
//     fp32 MFMA            v_mfma_f32_32x32x2_f32          (what the library uses today)
//     bf16 x 3, 6 products  a0b0 + a0b1 + a1b0 + a0b2 + a1b1 + a2b0 on v_mfma_f32_32x32x16_bf16
//     bf16 x 3, 3 products  a0b0 + a0b1 + a1b0              (for scale: ~16 good bits)
// Not product code; nothing in the library calls it. Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/bf16x3_accuracy.hip -o build_abl/bf16x3_accuracy && build_abl/bf16x3_accuracy
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ __bf16 to_bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    unsigned short h = (unsigned short)(u >> 16);
    return *reinterpret_cast<__bf16*>(&h);
}
__device__ __forceinline__ float bf16_to_f32(__bf16 b) {
    unsigned short h = *reinterpret_cast<unsigned short*>(&b);
    return __uint_as_float((unsigned)h << 16);
}

__global__ void split3(const float* x, __bf16* x0, __bf16* x1, __bf16* x2, long n) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    const __bf16 b0 = to_bf16_rne(v);
    const float r1 = v - bf16_to_f32(b0);
    const __bf16 b1 = to_bf16_rne(r1);
    const float r2 = r1 - bf16_to_f32(b1);
    x0[i] = b0; x1[i] = b1; x2[i] = to_bf16_rne(r2);
}

// one wave per 32 x 32 tile of C; fragments straight from global memory (an accuracy experiment, not a fast GEMM)
__global__ __launch_bounds__(64) void gemm_f32(const float* A, const float* B, float* C, int M, int N, int K) {
    const int lane = threadIdx.x, c = lane & 31, hh = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    f32x16 acc = {};
    for (int k = 0; k < K; k += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(long)(m0 + c) * K + k + hh], B[(long)(n0 + c) * K + k + hh], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[(long)(m0 + (r & 3) + 8 * (r >> 2) + 4 * hh) * N + n0 + c] = acc[r];
}

template <int NPROD>
__global__ __launch_bounds__(64) void gemm_bf16x3(const __bf16* A0, const __bf16* A1, const __bf16* A2, const __bf16* B0, const __bf16* B1,
                                                  const __bf16* B2, float* C, int M, int N, int K) {
    const int lane = threadIdx.x, c = lane & 31, hh = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    f32x16 acc = {};
    for (int k = 0; k < K; k += 16) {
        const long ia = (long)(m0 + c) * K + k + 8 * hh, ib = (long)(n0 + c) * K + k + 8 * hh;
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(A0 + ia), a1 = *reinterpret_cast<const bf16x8*>(A1 + ia), a2 = *reinterpret_cast<const bf16x8*>(A2 + ia);
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(B0 + ib), b1 = *reinterpret_cast<const bf16x8*>(B1 + ib), b2 = *reinterpret_cast<const bf16x8*>(B2 + ib);
        // smallest terms first
        if (NPROD == 6) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) C[(long)(m0 + (r & 3) + 8 * (r >> 2) + 4 * hh) * N + n0 + c] = acc[r];
}

static void report(const char* name, const std::vector<float>& c, const std::vector<double>& ref, const std::vector<double>& absref) {
    double max_rel = 0, sum_rel = 0;
    for (size_t i = 0; i < c.size(); ++i) {
        const double rel = fabs((double)c[i] - ref[i]) / absref[i];   // error relative to sum |a_k b_k|: the natural scale of a dot product
        max_rel = fmax(max_rel, rel); sum_rel += rel;
    }
    printf("%-28s max |err| / sum|a b| = %.3e   mean = %.3e\n", name, max_rel, sum_rel / c.size());
}

int main() {
    const int M = 512, N = 512, K = 256;
    for (int mode = 0; mode < 2; ++mode) {
        std::vector<float> a((size_t)M * K), b((size_t)N * K);
        srand(1 + mode);
        for (auto& v : a) { float u = rand() / (float)RAND_MAX - 0.5f; v = mode ? u * expf(8.f * (rand() / (float)RAND_MAX - 0.5f)) : u; }
        for (auto& v : b) { float u = rand() / (float)RAND_MAX - 0.5f; v = mode ? u * expf(8.f * (rand() / (float)RAND_MAX - 0.5f)) : u; }
        std::vector<double> ref((size_t)M * N), absref((size_t)M * N);
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < N; ++j) {
                double s = 0, t = 0;
                for (int k = 0; k < K; ++k) { const double p = (double)a[(size_t)i * K + k] * b[(size_t)j * K + k]; s += p; t += fabs(p); }
                ref[(size_t)i * N + j] = s; absref[(size_t)i * N + j] = t;
            }
        float *dA, *dB, *dC; __bf16 *a0, *a1, *a2, *b0, *b1, *b2;
        hipMalloc(&dA, a.size() * 4); hipMalloc(&dB, b.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
        hipMalloc(&a0, a.size() * 2); hipMalloc(&a1, a.size() * 2); hipMalloc(&a2, a.size() * 2);
        hipMalloc(&b0, b.size() * 2); hipMalloc(&b1, b.size() * 2); hipMalloc(&b2, b.size() * 2);
        hipMemcpy(dA, a.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, b.data(), b.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(split3, dim3((a.size() + 255) / 256), dim3(256), 0, 0, dA, a0, a1, a2, (long)a.size());
        hipLaunchKernelGGL(split3, dim3((b.size() + 255) / 256), dim3(256), 0, 0, dB, b0, b1, b2, (long)b.size());
        std::vector<float> c((size_t)M * N);
        printf("%s operands, K = %d\n", mode ? "wide dynamic range (e^-4 .. e^4)" : "uniform [-0.5, 0.5]", K);
        hipLaunchKernelGGL(gemm_f32, dim3(N / 32, M / 32), dim3(64), 0, 0, dA, dB, dC, M, N, K);
        hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost); report("fp32 MFMA", c, ref, absref);
        hipLaunchKernelGGL(gemm_bf16x3<6>, dim3(N / 32, M / 32), dim3(64), 0, 0, a0, a1, a2, b0, b1, b2, dC, M, N, K);
        hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost); report("bf16 x 3, 6 products", c, ref, absref);
        hipLaunchKernelGGL(gemm_bf16x3<3>, dim3(N / 32, M / 32), dim3(64), 0, 0, a0, a1, a2, b0, b1, b2, dC, M, N, K);
        hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost); report("bf16 x 3, 3 products", c, ref, absref);
    }
    return 0;
}
