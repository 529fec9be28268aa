// Micro-benchmark: what the fp32 matrix pipe of an MI355X sustains for v_mfma_f32_32x32x2_f32, alone and with the
// side work an attention step issues next to it (LDS fragment reads, exp2 VALU work). Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o build_abl/mfma_peak && build_abl/mfma_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: MFMA only; 1: + one ds_read_b128 per 4 MFMAs; 2: + one exp2 per 4 MFMAs; 3: both
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 68];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 64 * 68; i += 256) lds[i] = seed * i;
    __syncthreads();
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float x = seed * lane, y = seed + lane, e = seed;
    float4 f = *reinterpret_cast<const float4*>(&lds[(lane & 31) * 68]);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float4 g = f;
            if (MODE & 1) g = *reinterpret_cast<const float4*>(&lds[(lane & 31) * 68 + ((u + it) & 7) * 4 + (lane >> 5) * 32]);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(f.x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(f.y, y, a1, 0, 0, 0);
            if (MODE & 2) e = __builtin_amdgcn_exp2f(e - x);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(f.z, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w, x, a3, 0, 0, 0);
            f = g;
        }
    }
    float s = e;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
static void run(const char* name, int blocks_per_cu, float* out) {
    const int iters = 4000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int j = 0; j < 10; ++j) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1e-9f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 10.0 * grid * 4 * iters * 32.0 * 4096;
        if (rep == 2) printf("%-34s %d waves/SIMD: %7.1f TFLOP/s  (%.1f%% of 157.3)\n", name, blocks_per_cu, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100);
    }
}

// NV plain VALU fmas (NT transcendental exp2) per group of 4 MFMAs: how much of the matrix pipe does co-issued VALU work cost?
template <int NV, int NT_>
__global__ __launch_bounds__(256) void kv(float* out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float x = seed * lane, y = seed + lane;
    float v[16], e[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i] = seed * i; e[i] = seed + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i % 16] = __builtin_fmaf(v[i % 16], x, y);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NT_; ++i) e[i % 16] = __builtin_amdgcn_exp2f(e[i % 16]);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a3, 0, 0, 0);
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r] + v[r] + e[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int NT_>
static void runv(int blocks_per_cu, float* out) {
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int j = 0; j < 10; ++j) hipLaunchKernelGGL((kv<NV, NT_>), dim3(grid), dim3(256), 0, 0, out, iters, 1e-9f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 10.0 * grid * 4 * iters * 32.0 * 4096;
        if (rep == 1) printf("per 4 MFMA: %2d fma + %2d exp2, %d waves/SIMD: %7.1f TFLOP/s  (%.1f%%)\n", NV, NT_, blocks_per_cu, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100);
    }
}

// NC independent accumulator chains per wave (NC = 1: every MFMA waits for the previous one's result)
template <int NC>
__global__ __launch_bounds__(256) void kc(float* out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    f32x16 a[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) a[i] = f32x16{};
    float x = seed * lane, y = seed + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32 / NC; ++u)
#pragma unroll
            for (int i = 0; i < NC; ++i) a[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NC; ++i)
        for (int r = 0; r < 16; ++r) s += a[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NC>
static void runc(int blocks_per_cu, float* out) {
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int j = 0; j < 10; ++j) hipLaunchKernelGGL((kc<NC>), dim3(grid), dim3(256), 0, 0, out, iters, 1e-9f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 10.0 * grid * 4 * iters * 32.0 * 4096;
        if (rep == 1) printf("%d dependent chain(s) per wave, %d waves/SIMD: %7.1f TFLOP/s  (%.1f%%)\n", NC, blocks_per_cu, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100);
    }
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    // settle clocks
    for (int j = 0; j < 300; ++j) hipLaunchKernelGGL(k<0>, dim3(512), dim3(256), 0, 0, out, 4000, 1e-9f);
    hipDeviceSynchronize();
    for (int w = 1; w <= 2; ++w) {
        run<0>("mfma only", w, out);
        run<1>("mfma + ds_read_b128 / 4", w, out);
        run<2>("mfma + exp2 / 4", w, out);
        run<3>("mfma + ds_read_b128 + exp2 / 4", w, out);
    }
    for (int w = 1; w <= 2; ++w) {
        runv<0, 0>(w, out); runv<1, 0>(w, out); runv<2, 0>(w, out); runv<4, 0>(w, out); runv<8, 0>(w, out); runv<16, 0>(w, out); runv<32, 0>(w, out);
        runv<0, 1>(w, out); runv<0, 2>(w, out); runv<0, 4>(w, out); runv<0, 8>(w, out);
    }
    for (int w = 1; w <= 4; w *= 2) { runc<1>(w, out); runc<2>(w, out); runc<4>(w, out); }
    return 0;
}
