// Micro-benchmark: how fast can a CU of an MI355X fill LDS from L2, by the paths the Winograd convolution could use?
//   hipcc -O3 --offload-arch=gfx950 tools/dma_rate.hip -o build_abl/dma_rate && build_abl/dma_rate
// Every block re-reads the same 32 KB (an L2 / L1 resident working set, like the U block of a convolution slab, which every block
// of a layer reads) or its own patch-like window with a 256-byte stride between lanes, 8 "pieces" per wave and step:
//   MODE 0  buffer_load_dwordx4 ... lds            (LDS-DMA, 64 lanes x 16 B per piece, the convolution's form)
//   MODE 1  buffer_load_dwordx4 -> VGPR, ds_write_b128   (register staging)
//   MODE 2  buffer_load_dword ... lds              (LDS-DMA, 4 B per lane)
//   MODE 3  LDS-DMA x4 with a 256-byte lane stride (one pixel's 32-byte channel slab per lane pair: the patch transfer)
// and, for the question "does the transfer take matrix-pipe time?", MODE 0 with 32 MFMAs per step next to it (MODE 4).
// Reported: bytes per clock and CU (clock from wall time at the nominal 2.4 GHz) and ns per piece.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ u32x4 make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    u32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) & 0xFFFFu;
    r.z = __builtin_amdgcn_readfirstlane(bytes);
    r.w = 0x00020000u;
    return r;
}
__device__ __forceinline__ void dma16(u32x4 rsrc, unsigned lds_byte_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_byte_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma4(u32x4 rsrc, unsigned lds_byte_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(lds_byte_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ src, float* out, int iters, int src_bytes) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 x 32 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32x4 rs = make_rsrc(src, (unsigned)src_bytes);
    const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)lds;
    const unsigned voff = MODE == 3 ? (unsigned)(tid * 256 + (blockIdx.x & 63) * 65536) % (unsigned)(src_bytes - 4096) : (unsigned)tid * 16u;
    f32x16 acc[4] = {};
    float s = 0.f;
    for (int it = 0; it < iters; ++it) {
        const unsigned stage = lds0 + (it & 1) * 32768u + wave * 1024u;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            if (MODE == 0 || MODE == 3 || MODE == 4) dma16(rs, stage + p * 4096u, voff, p * 4096u);
            else if (MODE == 2) dma4(rs, stage + p * 4096u, voff, p * 4096u);
            else {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00020000), voff, p * 4096u, 0);
                *reinterpret_cast<u32x4*>(lds + (it & 1) * 8192 + p * 1024 + tid * 4) = v;
            }
        }
        if (MODE == 4) {
            const float x = lds[lane], y = lds[64 + lane];
#pragma unroll
            for (int j = 0; j < 32; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[j & 3], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        s += lds[(it & 1) * 8192 + tid];
    }
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
static void run(const char* name, int blocks_per_cu, const float* src, float* out, int src_bytes) {
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int j = 0; j < 5; ++j) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 65536, 0, src, out, iters, src_bytes);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double pieces_per_cu = 5.0 * blocks_per_cu * 4 * 8 * iters;              // per CU
        const double bytes_per_piece = MODE == 2 ? 256.0 : 1024.0;
        const double clk = ms * 1e-3 * 2.4e9;
        if (rep == 2)
            printf("%-58s %d block(s)/CU: %6.1f B/clk/CU  %6.1f clk per piece per CU  (%.2f TB/s over 256 CUs)%s\n", name, blocks_per_cu,
                   pieces_per_cu * bytes_per_piece / clk, clk / pieces_per_cu, pieces_per_cu * bytes_per_piece * 256 / (ms * 1e-3) / 1e12,
                   MODE == 4 ? "" : "");
        if (rep == 2 && MODE == 4) {
            const double fl = 5.0 * grid * 4 * iters * 32.0 * 4096;
            printf("%-58s          MFMA next to it: %.1f TFLOP/s (%.1f%% of 157.3)\n", "", fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100);
        }
    }
}

int main() {
    const int src_bytes = 64 << 20;
    float *src, *out;
    hipMalloc(&src, src_bytes);
    hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
    hipMemset(src, 0, src_bytes);
    for (int j = 0; j < 100; ++j) hipLaunchKernelGGL(k<0>, dim3(512), dim3(256), 65536, 0, src, out, 2000, src_bytes);
    hipDeviceSynchronize();
    for (int w = 1; w <= 2; ++w) {
        run<0>("LDS-DMA dwordx4, 32 KB shared by all blocks", w, src, out, src_bytes);
        run<1>("dwordx4 to registers + ds_write_b128", w, src, out, src_bytes);
        run<2>("LDS-DMA dword (4 B per lane)", w, src, out, src_bytes);
        run<3>("LDS-DMA dwordx4, 256 B lane stride (patch-like)", w, src, out, src_bytes);
        run<4>("LDS-DMA dwordx4 + 32 MFMAs per 8 pieces", w, src, out, src_bytes);
    }
    return 0;
}
