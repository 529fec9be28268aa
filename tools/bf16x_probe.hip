// Probe (round 5): can the bf16 matrix cores of an MI355X stand in for its f32-input MFMA, which runs at the VECTOR rate (157 TFLOP/s)?
// An f32 value is the exact sum of three bf16 values (8 + 8 + 8 significant bits, round to nearest at every cut), so an f32 product
// a.b is the sum of up to nine bf16 products that the matrix core forms exactly and adds in f32. Two questions, both for the hardware:
//   (1) accuracy: error of C = A.B^T (32 x 32, K = 64 .. 4096) against an f64 sum, in units of 2^-24 . sum |a||b|, for
//       the f32 MFMA chain (what the library runs today), 3 / 6 / 9 bf16 products, and orderings of the six;
//   (2) rate: six v_mfma_f32_32x32x16_bf16 per 16 k against eight v_mfma_f32_32x32x2_f32 (512 cycles), with vector instructions between.
// Build + run on the GPU box:   hipcc -O3 --offload-arch=gfx950 tools/bf16x_probe.hip -o /tmp/bf16x_probe && /tmp/bf16x_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline float bf_round(float x) {      // x rounded to bf16 (nearest even: v_cvt_pk_bf16_f32), as a float
    __bf16 b = (__bf16)x;
    unsigned short u = __builtin_bit_cast(unsigned short, b);
    return __uint_as_float((unsigned)u << 16);
}
__device__ inline __bf16 as_bf(float x_exact) {  // a float that IS a bf16 value
    unsigned short u = (unsigned short)(__float_as_uint(x_exact) >> 16);
    return __builtin_bit_cast(__bf16, u);
}
struct Split { bf16x8 p[3]; };
__device__ inline Split split8(const float* x) {
    Split s;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float h = bf_round(x[j]);
        const float r1 = x[j] - h;
        const float m = bf_round(r1);
        const float r2 = r1 - m;
        const float l = bf_round(r2);
        s.p[0][j] = as_bf(h); s.p[1][j] = as_bf(m); s.p[2][j] = as_bf(l);
    }
    return s;
}

// MODE 0: f32 MFMA chain. 1: bf16 x3 (00 01 10). 2: x6, small terms first, one accumulator. 3: x9. 4: x6, 00 in one accumulator,
// the five corrections in a second one, added at the end. 5: x6 as 2, but every 64 k into a fresh accumulator that is added to the
// running sum by a vector add (what a flash-attention step would do). 6: x6, large term first.
template <int MODE>
__global__ __launch_bounds__(64) void acc_kernel(const float* A, const float* B, float* C, int K) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc = {}, acc2 = {};
    if (MODE == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], B[r * K + k + h], acc, 0, 0, 0);
    } else {
        f32x16 run = {};
        for (int k = 0; k < K; k += 16) {
            const Split a = split8(A + r * K + k + 8 * h), b = split8(B + r * K + k + 8 * h);
#define MM(i, j, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[i], b.p[j], c, 0, 0, 0)
            if (MODE == 1) { MM(0, 1, acc); MM(1, 0, acc); MM(0, 0, acc); }
            if (MODE == 2 || MODE == 5) { MM(0, 2, acc); MM(2, 0, acc); MM(1, 1, acc); MM(0, 1, acc); MM(1, 0, acc); MM(0, 0, acc); }
            if (MODE == 3) { MM(2, 2, acc); MM(1, 2, acc); MM(2, 1, acc); MM(0, 2, acc); MM(2, 0, acc); MM(1, 1, acc); MM(0, 1, acc); MM(1, 0, acc); MM(0, 0, acc); }
            if (MODE == 4) { MM(0, 2, acc2); MM(2, 0, acc2); MM(1, 1, acc2); MM(0, 1, acc2); MM(1, 0, acc2); MM(0, 0, acc); }
            if (MODE == 6) { MM(0, 0, acc); MM(0, 1, acc); MM(1, 0, acc); MM(1, 1, acc); MM(0, 2, acc); MM(2, 0, acc); }
#undef MM
            if (MODE == 5 && (k & 63) == 48) {
                for (int i = 0; i < 16; ++i) run[i] += acc[i];
                acc = f32x16{};
            }
        }
        if (MODE == 4) for (int i = 0; i < 16; ++i) acc[i] += acc2[i];
        if (MODE == 5) for (int i = 0; i < 16; ++i) acc[i] += run[i];
    }
    for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

template <int MODE>
static void accuracy(const char* name, int K, int dist, float* dA, float* dB, float* dC) {
    std::mt19937 rng(1234 + K + dist);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::uniform_real_distribution<float> ud(0.f, 1.f);
    std::vector<float> A(32 * K), B(32 * K), C(32 * 32);
    for (auto& v : A) v = dist == 0 ? nd(rng) : ud(rng);
    for (auto& v : B) v = dist == 0 ? nd(rng) : dist == 1 ? ud(rng) : nd(rng);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(acc_kernel<MODE>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double mx = 0, sq = 0, bias = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double ex = 0, sc = 0;
            for (int k = 0; k < K; ++k) { ex += (double)A[i * K + k] * B[j * K + k]; sc += std::fabs((double)A[i * K + k] * B[j * K + k]); }
            const double e = ((double)C[i * 32 + j] - ex) / sc * 16777216.0;
            mx = std::fmax(mx, std::fabs(e)); sq += e * e; bias += e;
        }
    printf("  %-44s K %5d %-9s max %8.3f  rms %8.3f  mean %+8.3f   [2^-24 sum|ab|]\n", name, K,
           dist == 0 ? "N x N" : dist == 1 ? "U x U" : "U x N", mx, std::sqrt(sq / 1024), bias / 1024);
}

// rate: per iteration NM bf16 MFMAs (32x32x16) on NACC accumulators with NV independent vector fmas after each MFMA
template <int NACC, int NV, int KIND>   // KIND 0: v_fma_f32 fillers; 1: the split sequence (cvt_pk + shifts + subs) as the filler
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x16{};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed * (lane + j)); b[j] = (__bf16)(seed + j); }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = seed * (lane + j);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u % NACC], 0, 0, 0);
            if (KIND == 0) {
#pragma unroll
                for (int j = 0; j < NV; ++j) v[j % 8] = __builtin_fmaf(v[j % 8], seed, 1.0f);
            } else {
#pragma unroll
                for (int j = 0; j < NV; ++j) { const float hh = bf_round(v[j % 8]); v[j % 8] = (v[j % 8] - hh) * 256.f + seed; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int NV, int KIND>
static void rate(int blocks_per_cu, float* out) {
    const int iters = 4000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int j = 0; j < 10; ++j) hipLaunchKernelGGL((rate_kernel<NACC, NV, KIND>), dim3(grid), dim3(256), 0, 0, out, iters, 1e-9f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double mfmas = 10.0 * grid * 4 * iters * 6.0;
    const double bf = mfmas * 2 * 32 * 32 * 16 / ms / 1e9;
    printf("  %d accumulators, %d %s per MFMA, %d waves/SIMD: %7.1f TFLOP/s bf16 = %6.1f f32-equivalent at six products (%.2f x 157.3)\n",
           NACC, NV, KIND ? "split steps (cvt_pk, sub, fma)" : "v_fma_f32", blocks_per_cu, bf, bf / 6, bf / 6 / 157.3);
}

// the two bf16 MFMA shapes on RANDOM operands (the clock the chip holds depends on the data: guide, DVFS give-back): same FLOPs per wave and iteration
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int SHAPE>   // 0: 32x32x16 (four accumulators of 16), 1: 16x16x32 (eight accumulators of 4: the same 32 x 32 x 64 of work per eight / sixteen MFMAs)
__global__ __launch_bounds__(256) void shape_kernel(float* out, int iters, unsigned seed) {
    unsigned h = (threadIdx.x + 1) * 2654435761u ^ (blockIdx.x * 40503u) ^ seed;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) {
            h = h * 1664525u + 1013904223u; a[i][j] = (__bf16)(((int)(h >> 8) & 0xffff) * (1.f / 32768.f) - 1.f);
            h = h * 1664525u + 1013904223u; b[i][j] = (__bf16)(((int)(h >> 8) & 0xffff) * (1.f / 32768.f) - 1.f);
        }
    float s = 0;
    if (SHAPE == 0) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = f32x16{};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u & 3], b[(u >> 1) & 3], acc[u & 3], 0, 0, 0);
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        f32x4v acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4v{};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u & 3], b[(u >> 2) & 3], acc[u & 7], 0, 0, 0);
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int SHAPE>
static void shape_rate(int blocks_per_cu, float* out) {
    const int iters = 4000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int j = 0; j < 25; ++j) hipLaunchKernelGGL((shape_kernel<SHAPE>), dim3(grid), dim3(256), 0, 0, out, iters, 77u + rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double fl = 25.0 * grid * 4 * iters * 8.0 * 2 * 32 * 32 * 16;
    printf("  %s on random operands, %d waves/SIMD: %7.1f TFLOP/s bf16 (%.0f ms of back-to-back launches)\n", SHAPE ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_32x32x16_bf16",
           blocks_per_cu, fl / ms / 1e9, ms);
}

int main() {
    float *dA, *dB, *dC, *out;
    hipMalloc(&dA, 32 * 4096 * 4); hipMalloc(&dB, 32 * 4096 * 4); hipMalloc(&dC, 32 * 32 * 4); hipMalloc(&out, 256 * 8 * 256 * 4);
    printf("accuracy of C = A.B^T (32 x 32) against f64, error / (2^-24 sum |a||b|):\n");
    for (int dist = 0; dist < 3; ++dist)
        for (int K : {64, 256, 4096}) {
            accuracy<0>("f32 MFMA chain (32x32x2)", K, dist, dA, dB, dC);
            accuracy<1>("bf16 x3 (00 01 10)", K, dist, dA, dB, dC);
            accuracy<2>("bf16 x6, small terms first", K, dist, dA, dB, dC);
            accuracy<6>("bf16 x6, large term first", K, dist, dA, dB, dC);
            accuracy<4>("bf16 x6, corrections in a 2nd accumulator", K, dist, dA, dB, dC);
            accuracy<5>("bf16 x6, fresh accumulator per 64 k + v_add", K, dist, dA, dB, dC);
            accuracy<3>("bf16 x9", K, dist, dA, dB, dC);
        }
    printf("rate of six v_mfma_f32_32x32x16_bf16 per 16 k:\n");
    rate<1, 0, 0>(1, out); rate<2, 0, 0>(1, out); rate<4, 0, 0>(1, out); rate<4, 0, 0>(2, out);
    rate<4, 2, 0>(1, out); rate<4, 4, 0>(1, out); rate<4, 5, 0>(1, out); rate<4, 6, 0>(1, out); rate<4, 8, 0>(1, out);
    rate<4, 4, 0>(2, out); rate<4, 6, 0>(2, out); rate<4, 8, 0>(2, out);
    rate<4, 1, 1>(1, out); rate<4, 2, 1>(1, out); rate<4, 2, 1>(2, out);
    printf("the two bf16 shapes on random operands (same FLOPs per iteration):\n");
    shape_rate<0>(1, out); shape_rate<1>(1, out); shape_rate<0>(2, out); shape_rate<1>(2, out);
    return 0;
}
