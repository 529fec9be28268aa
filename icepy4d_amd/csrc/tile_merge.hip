// Tile-mode merge on the device (row f-1 of the scope table): the tail of `ImageMatcherBase._match_by_tile`
// (`src/icepy4d/matching/matchers.py:402-448`) for ALL tile pairs of an image pair at once:
//   for every tile pair p (in the order the reference loops over them) and every keypoint i of its first tile with a match:
//       mkpts0 = (kpts0[i] + tile origin) + image origin,   mkpts1 likewise          (two fp32 additions, the reference's order)
//   then `np.unique(mkpts0_full, axis=0, return_index=True)`: rows sorted lexicographically (x, then y), first occurrence kept.
// No sort: duplicates are found and the unique rows are ordered by counting (rank_sweep.h); the number of matched points is a
// few thousand to a few ten thousand, far below the P x K capacity the grids are sized for.
#include "common.h"
#include "ctx.h"
#include "lg_misc.h"
#include "rank_sweep.h"

namespace im {

__global__ __launch_bounds__(256) void tm_collect_kernel(int K, const int* __restrict__ matches, const int* __restrict__ slots,
                                                          const float* __restrict__ off, float4 org, const float* __restrict__ kp_bank,
                                                          const int* __restrict__ n_bank, unsigned long long* __restrict__ key,
                                                          unsigned* __restrict__ seq, int* __restrict__ count) {
    __shared__ int cnt[2];
    const int p = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int t0 = slots[2 * p];
    if (threadIdx.x == 0) cnt[0] = 0;
    __syncthreads();
    bool take = false;
    unsigned long long k = 0;
    if (i < K && i < n_bank[t0] && matches[(long)p * K + i] > -1) {
        const float x = (kp_bank[((long)t0 * K + i) * 2] + off[4 * p]) + org.x;
        const float y = (kp_bank[((long)t0 * K + i) * 2 + 1] + off[4 * p + 1]) + org.y;
        k = ((unsigned long long)f2ord(x) << 32) | f2ord(y);
        take = true;
    }
    const int local = take ? atomicAdd(&cnt[0], 1) : 0;
    __syncthreads();
    if (threadIdx.x == 0 && cnt[0]) cnt[1] = atomicAdd(&count[0], cnt[0]);
    __syncthreads();
    if (take) {
        key[cnt[1] + local] = k;
        seq[cnt[1] + local] = (unsigned)(p * K + i);
    }
}

// later duplicates of a row get the largest key, so they rank behind every first occurrence
__global__ __launch_bounds__(RK_N) void tm_first_kernel(const unsigned long long* __restrict__ key, const unsigned* __restrict__ seq,
                                                         const int* __restrict__ count, unsigned long long* __restrict__ skey) {
    __shared__ __attribute__((aligned(16))) unsigned long long tile[1024];
    __shared__ unsigned stile[1024];
    __shared__ int part[RK_N];
    const int m = count[0];
    if (blockIdx.x * RK_T >= m) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * RK_T + lane;
    const unsigned long long me = i < m ? key[i] : 0ull;
    const unsigned myseq = i < m ? seq[i] : 0u;
    int dup = 0;
    for (int base = 0; base < m; base += 1024) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 1024 / RK_N; ++j) {
            const int e = base + j * RK_N + threadIdx.x;
            tile[j * RK_N + threadIdx.x] = e < m ? key[e] : 0ull;
            stile[j * RK_N + threadIdx.x] = e < m ? seq[e] : 0xFFFFFFFFu;
        }
        __syncthreads();
        for (int e = wave * (1024 / RK_W); e < (wave + 1) * (1024 / RK_W); ++e) dup += (tile[e] == me) & (stile[e] < myseq);
    }
    __syncthreads();
    part[wave * RK_T + lane] = dup;
    __syncthreads();
    if (threadIdx.x < RK_T && i < m) {
        int total = 0;
#pragma unroll
        for (int w = 0; w < RK_W; ++w) total += part[w * RK_T + lane];
        skey[i] = total ? ~0ull : me;
    }
}

__global__ __launch_bounds__(RK_N) void tm_rank_kernel(const unsigned long long* __restrict__ skey, const unsigned* __restrict__ seq,
                                                        int* __restrict__ count, int K, const int* __restrict__ matches,
                                                        const int* __restrict__ slots, const float* __restrict__ off, float4 org,
                                                        const float* __restrict__ kp_bank, int* __restrict__ idx0, int* __restrict__ idx1,
                                                        float* __restrict__ kp0, float* __restrict__ kp1) {
    __shared__ __attribute__((aligned(16))) unsigned long long tile[1024];
    __shared__ int part[RK_N];
    const int m = count[0];
    if (blockIdx.x * RK_T >= m) return;
    const int i = blockIdx.x * RK_T + (threadIdx.x & 63);
    const unsigned long long mine = i < m ? skey[i] : ~0ull;
    // rank_among counts LARGER keys; ascending order wanted: complement the keys (first occurrences stay below ~0 -> above 0)
    const int larger = rank_among_inv(skey, m, mine, tile, part);
    if (threadIdx.x < RK_T && i < m && mine != ~0ull) {
        const int r = larger;                                  // number of first-occurrence keys smaller than mine
        const unsigned sq = seq[i];
        const int p = (int)(sq / (unsigned)K), row = (int)(sq - (unsigned)p * K);
        const int t0 = slots[2 * p], t1 = slots[2 * p + 1];
        const int j = matches[(long)p * K + row];
        idx0[r] = t0 * K + row;
        idx1[r] = t1 * K + j;
        kp0[2 * r] = (kp_bank[((long)t0 * K + row) * 2] + off[4 * p]) + org.x;
        kp0[2 * r + 1] = (kp_bank[((long)t0 * K + row) * 2 + 1] + off[4 * p + 1]) + org.y;
        kp1[2 * r] = (kp_bank[((long)t1 * K + j) * 2] + off[4 * p + 2]) + org.z;
        kp1[2 * r + 1] = (kp_bank[((long)t1 * K + j) * 2 + 1] + off[4 * p + 3]) + org.w;
        atomicAdd(&count[1], 1);
    }
}

// dst[r][:] = src[idx[r]][:], rows of `row_floats` floats (a multiple of 4 uses 16-byte moves)
__global__ __launch_bounds__(256) void gather_rows_generic_kernel(const float* __restrict__ src, int row_floats, const int* __restrict__ idx,
                                                                   int n, float* __restrict__ dst) {
    const int lanes = row_floats >= 256 ? 64 : (row_floats >= 4 ? max(row_floats / 4, 1) : 1);
    const int per_block = 256 / lanes;
    const int r = blockIdx.x * per_block + threadIdx.x / lanes, l = threadIdx.x % lanes;
    if (r >= n || threadIdx.x / lanes >= per_block) return;
    const float* s = src + (long)idx[r] * row_floats;
    float* d = dst + (long)r * row_floats;
    if ((row_floats & 3) == 0)
        for (int c = l * 4; c < row_floats; c += lanes * 4) *reinterpret_cast<float4*>(d + c) = *reinterpret_cast<const float4*>(s + c);
    else
        for (int c = l; c < row_floats; c += lanes) d[c] = s[c];
}

}  // namespace im

using namespace im;

extern "C" {

int im_merge_tile_matches(im_ctx* ctx, int n_pairs, int max_kpts, const int32_t* d_matches, const int32_t* d_slots, const float* d_off,
                          const float* h_origin, const float* d_kp_bank, const int32_t* d_n_bank, int32_t* d_count, int32_t* d_idx0,
                          int32_t* d_idx1, float* d_kp0, float* d_kp1, void* stream) {
    IM_CHECK_CTX(ctx);
    if (n_pairs < 1 || max_kpts < 1 || !d_matches || !d_slots || !d_off || !h_origin || !d_kp_bank || !d_n_bank || !d_count)
        return ctx->fail(-70, "im_merge_tile_matches: bad arguments");
    const size_t cap = (size_t)n_pairs * max_kpts;
    if (!ctx->merge) ctx->merge = new MergeScratch();
    MergeScratch& ms = *ctx->merge;
    if (cap > ms.cap) {
        IM_HIP(ctx, hipDeviceSynchronize());
        ctx->dfree(ms.key); ctx->dfree(ms.skey); ctx->dfree(ms.seq);     // the smaller scratch this replaces
        ms.key = ms.skey = nullptr; ms.seq = nullptr; ms.cap = 0;
        ms.key = ctx->dalloc<unsigned long long>(cap, "merge.key");
        ms.skey = ctx->dalloc<unsigned long long>(cap, "merge.skey");
        ms.seq = ctx->dalloc<unsigned>(cap, "merge.seq");
        if (!ms.count) ms.count = ctx->dalloc<int>(4, "merge.count");
        if (!ms.key || !ms.skey || !ms.seq || !ms.count) return ctx->fail(-71, "im_merge_tile_matches: out of device memory");
        ms.cap = cap;
    }
    hipStream_t s = (hipStream_t)stream;
    const float4 org = make_float4(h_origin[0], h_origin[1], h_origin[2], h_origin[3]);
    IM_HIP(ctx, launch_zero_words(ms.count, 4, s));
    hipLaunchKernelGGL(tm_collect_kernel, dim3((max_kpts + 255) / 256, n_pairs), dim3(256), 0, s, max_kpts, d_matches, d_slots, d_off, org,
                       d_kp_bank, d_n_bank, ms.key, ms.seq, ms.count);
    const unsigned nb = (unsigned)((cap + RK_T - 1) / RK_T);
    hipLaunchKernelGGL(tm_first_kernel, dim3(nb), dim3(RK_N), 0, s, ms.key, ms.seq, ms.count, ms.skey);
    hipLaunchKernelGGL(tm_rank_kernel, dim3(nb), dim3(RK_N), 0, s, ms.skey, ms.seq, ms.count, max_kpts, d_matches, d_slots, d_off, org, d_kp_bank,
                       d_idx0, d_idx1, d_kp0, d_kp1);
    IM_HIP(ctx, hipGetLastError());
    IM_HIP(ctx, hipMemcpyAsync(d_count, ms.count + 1, sizeof(int), hipMemcpyDeviceToDevice, s));
    IM_GUARD_CHECK(ctx, s, "im_merge_tile_matches");
    return 0;
}

int im_gather_rows(im_ctx* ctx, const float* d_src, int row_floats, const int32_t* d_idx, int n, float* d_dst, void* stream) {
    IM_CHECK_CTX(ctx);
    if (n <= 0) return 0;
    if (row_floats < 1 || !d_src || !d_idx || !d_dst) return ctx->fail(-72, "im_gather_rows: bad arguments");
    const int lanes = row_floats >= 256 ? 64 : (row_floats >= 4 ? (row_floats / 4 > 0 ? row_floats / 4 : 1) : 1);
    const int per_block = 256 / lanes;
    hipLaunchKernelGGL(gather_rows_generic_kernel, dim3((n + per_block - 1) / per_block), dim3(256), 0, (hipStream_t)stream, d_src, row_floats, d_idx,
                       n, d_dst);
    IM_HIP(ctx, hipGetLastError());
    return 0;
}

}  // extern "C"
