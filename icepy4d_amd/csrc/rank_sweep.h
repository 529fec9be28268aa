// Rank by counting: the order of a few thousand unique 64-bit keys without a sort (shared by the keypoint top-k selection in
// sp_post.hip and the tile-match merge in tile_merge.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace im {

// rank of `mine` among src[0 .. m): number of keys larger under `mask`. A block ranks RK_T = 64 keys (lane l of every wave holds
// key l) and its RK_W waves each sweep their own slice of every 1024-key tile (LDS broadcast reads, two keys per read); the
// partial ranks meet in LDS. The sweep is bound by the 64-bit compares (one wave needs ~12 ns per key), so k keys are spread
// over k / 64 blocks x 8 waves.
static constexpr int RK_T = 64, RK_W = 8, RK_N = RK_T * RK_W;
__device__ __forceinline__ int rank_among(const unsigned long long* __restrict__ src, int m, unsigned long long mine,
                                          unsigned long long mask, unsigned long long* tile, int* part) {
    int rank = 0;
    const unsigned long long me = mine & mask;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int base = 0; base < m; base += 1024) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 1024 / RK_N; ++j) {
            const int i = base + j * RK_N + threadIdx.x;
            tile[j * RK_N + threadIdx.x] = i < m ? (src[i] & mask) : 0ull;      // 0 is smaller than every real key
        }
        __syncthreads();
        // the padding entries of a partial tile are 0 and never count: always sweep the whole slice, 8 reads in flight
        const ulonglong2* t2 = reinterpret_cast<const ulonglong2*>(tile) + wave * (512 / RK_W);
#pragma unroll 1
        for (int i = 0; i < 512 / RK_W; i += 8) {
            ulonglong2 kk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) kk[u] = t2[i + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) rank += (kk[u].x > me) + (kk[u].y > me);
        }
    }
    __syncthreads();
    part[wave * RK_T + lane] = rank;
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int w = 0; w < RK_W; ++w) total += part[w * RK_T + lane];
    return total;
}

// the same sweep counting SMALLER keys (ascending order), no mask
__device__ __forceinline__ int rank_among_inv(const unsigned long long* __restrict__ src, int m, unsigned long long mine,
                                              unsigned long long* tile, int* part) {
    int rank = 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int base = 0; base < m; base += 1024) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 1024 / RK_N; ++j) {
            const int i = base + j * RK_N + threadIdx.x;
            tile[j * RK_N + threadIdx.x] = i < m ? src[i] : ~0ull;             // ~0 is not smaller than any key
        }
        __syncthreads();
        const ulonglong2* t2 = reinterpret_cast<const ulonglong2*>(tile) + wave * (512 / RK_W);
#pragma unroll 1
        for (int i = 0; i < 512 / RK_W; i += 8) {
            ulonglong2 kk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) kk[u] = t2[i + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) rank += (kk[u].x < mine) + (kk[u].y < mine);
        }
    }
    __syncthreads();
    part[wave * RK_T + lane] = rank;
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int w = 0; w < RK_W; ++w) total += part[w * RK_T + lane];
    return total;
}

}  // namespace im
