// SuperPoint post-processing: detector softmax + depth-to-space, simple_nms, border/threshold/top-k
// selection, descriptor sampling. All HBM-bound integer / compare work: coalesced loads, LDS tiles for the
// stencils, wavefront reductions; no matrix cores.
#include "common.h"
#include "kernels.h"
#include "sp_post.h"

namespace im {

// ---------------------------------------------------------------------------------------------------------
// detector head tail (`lightglue/superpoint.py:170-173`): softmax over 65 logits per cell, drop the dustbin,
// channel c of cell (i, j) -> pixel (8i + c/8, 8j + c%8). One wave per cell.
__global__ __launch_bounds__(256) void det_softmax_shuffle_kernel(const float* __restrict__ logits, int ld,
                                                                   float* __restrict__ smap, int B, int hc, int wc) {
    const int lane = threadIdx.x & 63;
    const long cell = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long ncell = (long)B * hc * wc;
    if (cell >= ncell) return;
    const float* lp = logits + cell * ld;
    const float v = lp[lane];
    const float dust = lp[64];
    const float mx = fmaxf(wave_max(v), dust);
    const float e = expf(v - mx);
    const float sum = wave_sum(e) + expf(dust - mx);
    const int b = (int)(cell / ((long)hc * wc));
    const int rem = (int)(cell - (long)b * hc * wc);
    const int i = rem / wc, j = rem - i * wc;
    const int W8 = wc * 8;
    smap[((long)b * hc * 8 + 8 * i + (lane >> 3)) * W8 + 8 * j + (lane & 7)] = e / sum;
}

hipError_t launch_det_softmax(const float* logits, int ld, float* smap, int B, int hc, int wc, hipStream_t s) {
    const long ncell = (long)B * hc * wc;
    hipLaunchKernelGGL(det_softmax_shuffle_kernel, dim3((unsigned)((ncell + 3) / 4)), dim3(256), 0, s, logits, ld, smap, B, hc, wc);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// simple_nms (`lightglue/superpoint.py:50-65`): five (2r+1)^2 max-pools (stride 1, -inf padding) chained with
// exact fp32 equality tests. Each stage is one launch; the pooled quantity of a 32 x 64 tile plus halo r is
// staged in LDS and reduced separably (row max, then column max).
static constexpr int NT_H = 32, NT_W = 64, NMS_RMAX = 8;

template <typename LoadF>
__device__ __forceinline__ void pool_tile(LoadF load, int r, int ty0, int tx0, float* sT, float* sH, float out[8]) {
    const int tid = threadIdx.x;
    const int th = NT_H + 2 * r, tw = NT_W + 2 * r;
    for (int idx = tid; idx < th * tw; idx += 256) {
        const int y = idx / tw, x = idx - y * tw;
        sT[idx] = load(ty0 + y - r, tx0 + x - r);
    }
    __syncthreads();
    for (int idx = tid; idx < th * NT_W; idx += 256) {
        const int y = idx / NT_W, x = idx - y * NT_W;
        const float* p = sT + y * tw + x;
        float m = p[0];
        for (int d = 1; d <= 2 * r; ++d) m = fmaxf(m, p[d]);
        sH[idx] = m;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int p = tid + i * 256;
        const int y = p / NT_W, x = p - y * NT_W;
        const float* q = sH + y * NT_W + x;
        float m = q[0];
        for (int d = 1; d <= 2 * r; ++d) m = fmaxf(m, q[d * NT_W]);
        out[i] = m;
    }
}

// STAGE 0: mask = (s == pool(s))
// STAGE 1: supp = pool(mask) > 0 ; rest = supp ? 0 : s
// STAGE 2: mask |= (rest == pool(rest)) & !supp ; if FINAL: out = mask ? s : 0
template <int STAGE, bool FINAL>
__global__ __launch_bounds__(256) void nms_stage_kernel(const float* __restrict__ s, uint8_t* __restrict__ mask,
                                                        uint8_t* __restrict__ supp, float* __restrict__ rest,
                                                        float* __restrict__ out, int H, int W, int r) {
    __shared__ float sT[(NT_H + 2 * NMS_RMAX) * (NT_W + 2 * NMS_RMAX)];
    __shared__ float sH[(NT_H + 2 * NMS_RMAX) * NT_W];
    const long img = (long)blockIdx.z * H * W;
    const int ty0 = blockIdx.y * NT_H, tx0 = blockIdx.x * NT_W;
    float pooled[8];
    auto ld_f = [&](const float* src) {
        return [=](int y, int x) { return (y >= 0 && y < H && x >= 0 && x < W) ? src[img + (long)y * W + x] : -INFINITY; };
    };
    if constexpr (STAGE == 0) {
        pool_tile(ld_f(s), r, ty0, tx0, sT, sH, pooled);
    } else if constexpr (STAGE == 1) {
        auto ld_m = [=](int y, int x) { return (y >= 0 && y < H && x >= 0 && x < W) ? (float)mask[img + (long)y * W + x] : -INFINITY; };
        pool_tile(ld_m, r, ty0, tx0, sT, sH, pooled);
    } else {
        pool_tile(ld_f(rest), r, ty0, tx0, sT, sH, pooled);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int p = threadIdx.x + i * 256;
        const int y = ty0 + p / NT_W, x = tx0 + p % NT_W;
        if (y >= H || x >= W) continue;
        const long g = img + (long)y * W + x;
        if constexpr (STAGE == 0) {
            mask[g] = (s[g] == pooled[i]) ? 1 : 0;
        } else if constexpr (STAGE == 1) {
            const bool sp = pooled[i] > 0.f;
            supp[g] = sp ? 1 : 0;
            rest[g] = sp ? 0.f : s[g];
        } else {
            const bool m = mask[g] | ((rest[g] == pooled[i]) & (supp[g] == 0));
            if constexpr (FINAL) out[g] = m ? s[g] : 0.f;
            else mask[g] = m ? 1 : 0;
        }
    }
}

hipError_t launch_nms(const float* s, float* out, uint8_t* mask, uint8_t* supp, float* rest, int B, int H, int W, int r,
                      hipStream_t st) {
    if (r < 0 || r > NMS_RMAX) return hipErrorInvalidValue;
    dim3 grid((W + NT_W - 1) / NT_W, (H + NT_H - 1) / NT_H, B), block(256);
    hipLaunchKernelGGL((nms_stage_kernel<0, false>), grid, block, 0, st, s, mask, supp, rest, out, H, W, r);
    hipLaunchKernelGGL((nms_stage_kernel<1, false>), grid, block, 0, st, s, mask, supp, rest, out, H, W, r);
    hipLaunchKernelGGL((nms_stage_kernel<2, false>), grid, block, 0, st, s, mask, supp, rest, out, H, W, r);
    hipLaunchKernelGGL((nms_stage_kernel<1, false>), grid, block, 0, st, s, mask, supp, rest, out, H, W, r);
    hipLaunchKernelGGL((nms_stage_kernel<2, true>), grid, block, 0, st, s, mask, supp, rest, out, H, W, r);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// keypoint selection (`lightglue/superpoint.py:177-200`; SuperGlue flavour `models/superpoint.py:176-203`):
// candidates = pixels outside the border frame with score > threshold, in row-major order (torch.where /
// nonzero order). If there are more than k, keep the k largest, sorted descending. Key = score bits in the
// high word, ~flat-index in the low word: unique, and ties resolve to the lower flat index (torch.topk's tie
// order is unspecified).
static constexpr int SEL_CHUNK = 1024;

__device__ __forceinline__ bool is_cand(float v, long idx, int H, int W, int border, float thr) {
    const int y = (int)(idx / W), x = (int)(idx - (long)y * W);
    return (v > thr) && y >= border && y < H - border && x >= border && x < W - border;
}

__global__ __launch_bounds__(256) void kp_count_kernel(const float* __restrict__ nms, int H, int W, int border, float thr,
                                                        int* __restrict__ counts, int nchunks) {
    const int b = blockIdx.y;
    const long npix = (long)H * W;
    const float* s = nms + (long)b * npix;
    const long base = (long)blockIdx.x * SEL_CHUNK + threadIdx.x * 4;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long idx = base + j;
        if (idx < npix && is_cand(s[idx], idx, H, W, border, thr)) ++cnt;
    }
    __shared__ int red[4];
    cnt = wave_sum_i(cnt);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) counts[(long)b * nchunks + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// exclusive scan of the chunk counts of one image (single block), total -> n_cand[b]
__global__ __launch_bounds__(1024) void kp_scan_kernel(int* __restrict__ counts, int nchunks, int* __restrict__ n_cand) {
    const int b = blockIdx.x;
    int* c = counts + (long)b * nchunks;
    __shared__ int part[1024];
    const int per = (nchunks + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(lo + per, nchunks);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += c[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partials
    for (int off = 1; off < 1024; off <<= 1) {
        int v = (threadIdx.x >= off) ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;  // exclusive prefix of this thread's range
    for (int i = lo; i < hi; ++i) {
        const int v = c[i];
        c[i] = run;
        run += v;
    }
    if (threadIdx.x == 1023) n_cand[b] = part[1023];
}

__global__ __launch_bounds__(256) void kp_scatter_kernel(const float* __restrict__ nms, int H, int W, int border, float thr,
                                                          const int* __restrict__ offsets, int nchunks,
                                                          unsigned long long* __restrict__ keys, long key_stride) {
    const int b = blockIdx.y;
    const long npix = (long)H * W;
    const float* s = nms + (long)b * npix;
    const long base = (long)blockIdx.x * SEL_CHUNK + threadIdx.x * 4;
    float v[4];
    bool f[4];
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long idx = base + j;
        v[j] = idx < npix ? s[idx] : 0.f;
        f[j] = idx < npix && is_cand(v[j], idx, H, W, border, thr);
        cnt += f[j];
    }
    // block exclusive scan of cnt (row-major order = thread order)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    __shared__ int wtot[4];
    if (lane == 63) wtot[wv] = inc;
    __syncthreads();
    int pos = offsets[(long)b * nchunks + blockIdx.x] + inc - cnt;
    for (int i = 0; i < wv; ++i) pos += wtot[i];
    unsigned long long* kd = keys + (long)b * key_stride;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (f[j]) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(v[j]) << 32) |
                                           (unsigned long long)(0xFFFFFFFFu - (unsigned)(base + j));
            kd[pos++] = key;
        }
}

__device__ __forceinline__ void emit_kp(unsigned long long key, int W, float* kp, float* sc) {
    const unsigned idx = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
    const int y = idx / W, x = idx - y * W;
    kp[0] = (float)x;
    kp[1] = (float)y;
    *sc = __uint_as_float((unsigned)(key >> 32));
}

// one block per image: n <= k -> row-major copy; else radix-select the k-th largest key (8 passes of 8 bits),
// collect the k keys >= it into LDS, bitonic-sort them descending, emit.
__global__ __launch_bounds__(1024) void kp_topk_kernel(const unsigned long long* __restrict__ keys, long key_stride,
                                                        const int* __restrict__ n_cand, int k_req, int kmax, int W,
                                                        float* __restrict__ kpts, float* __restrict__ scores,
                                                        int* __restrict__ n_out, int pow2) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];  // [pow2]
    __shared__ int hist[256];
    __shared__ int sh_digit, sh_remaining, sh_count;
    const int b = blockIdx.x, tid = threadIdx.x;
    const unsigned long long* kd = keys + (long)b * key_stride;
    float* kp = kpts + (long)b * kmax * 2;
    float* sc = scores + (long)b * kmax;
    const int n = n_cand[b];
    const int k = (k_req > 0 && k_req < kmax) ? k_req : kmax;
    if (n <= k) {
        for (int i = tid; i < n; i += 1024) emit_kp(kd[i], W, kp + 2 * i, sc + i);
        if (tid == 0) n_out[b] = n;
        return;
    }
    unsigned long long prefix = 0;
    int remaining = k;
    for (int pass = 7; pass >= 0; --pass) {
        const int shift = pass * 8;
        for (int i = tid; i < 256; i += 1024) hist[i] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += 1024) {
            const unsigned long long key = kd[i];
            const bool match = (pass == 7) || ((key >> (shift + 8)) == (prefix >> (shift + 8)));
            if (match) atomicAdd(&hist[(int)((key >> shift) & 255ull)], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int cum = 0, d = 255;
            for (; d > 0; --d) {
                if (cum + hist[d] >= remaining) break;
                cum += hist[d];
            }
            sh_digit = d;
            sh_remaining = remaining - cum;
        }
        __syncthreads();
        prefix |= (unsigned long long)sh_digit << shift;
        remaining = sh_remaining;
        const int in_bin = hist[sh_digit];
        __syncthreads();
        // all 32 score bits are fixed after pass 4; if every key with exactly that score is wanted there is no tie at
        // the cut, the index passes cannot change anything and the threshold key is (score, lowest possible index word)
        if (pass == 4 && remaining == in_bin) break;
    }
    // prefix is now the k-th largest key; keys are unique so exactly k keys are >= prefix
    if (tid == 0) sh_count = 0;
    for (int i = tid; i < pow2; i += 1024) sk[i] = 0ull;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const unsigned long long key = kd[i];
        if (key >= prefix) {
            const int p = atomicAdd(&sh_count, 1);
            if (p < pow2) sk[p] = key;
        }
    }
    __syncthreads();
    for (int size = 2; size <= pow2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < (pow2 >> 1); i += 1024) {
                const int pos = 2 * i - (i & (stride - 1));
                const unsigned long long x = sk[pos], y = sk[pos + stride];
                const bool first_half = (pos & size) == 0;  // descending overall
                if ((x < y) == first_half) { sk[pos] = y; sk[pos + stride] = x; }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < k; i += 1024) emit_kp(sk[i], W, kp + 2 * i, sc + i);
    if (tid == 0) n_out[b] = k;
}

hipError_t launch_select_topk(const float* nms, int B, int H, int W, int border, float thr, int k_req, int kmax,
                              int* counts, int* n_cand, unsigned long long* keys, float* kpts, float* scores,
                              int* n_out, hipStream_t st) {
    const long npix = (long)H * W;
    const int nchunks = (int)((npix + SEL_CHUNK - 1) / SEL_CHUNK);
    hipLaunchKernelGGL(kp_count_kernel, dim3(nchunks, B), dim3(256), 0, st, nms, H, W, border, thr, counts, nchunks);
    hipLaunchKernelGGL(kp_scan_kernel, dim3(B), dim3(1024), 0, st, counts, nchunks, n_cand);
    hipLaunchKernelGGL(kp_scatter_kernel, dim3(nchunks, B), dim3(256), 0, st, nms, H, W, border, thr, counts, nchunks, keys, npix);
    const int k = (k_req > 0 && k_req < kmax) ? k_req : kmax;
    int pow2 = 2;
    while (pow2 < k) pow2 <<= 1;
    const size_t lds = (size_t)pow2 * sizeof(unsigned long long);
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&kp_topk_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_lds = lds;
    }
    hipLaunchKernelGGL(kp_topk_kernel, dim3(B), dim3(1024), lds, st, keys, npix, n_cand, k_req, kmax, W, kpts, scores, n_out, pow2);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// sample_descriptors (`lightglue/superpoint.py:75-87`) fused with the dense per-cell L2 normalisation
// (`:205`): one wave per keypoint, each lane 4 of the 256 channels; the four bilinear taps are whole 1 KiB
// rows of the NHWC descriptor map, normalised on the fly (x / max(||x||, 1e-12)), combined with the
// align_corners=True weights (zero padding outside the map) and normalised again.
__global__ __launch_bounds__(256) void sample_desc_kernel(const float* __restrict__ dense, int hc, int wc,
                                                           const float* __restrict__ kpts, const int* __restrict__ n_ptr,
                                                           int kmax, float* __restrict__ desc) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int kp = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (kp >= n_ptr[b]) return;
    const float kx = kpts[((long)b * kmax + kp) * 2], ky = kpts[((long)b * kmax + kp) * 2 + 1];
    float gx = ((kx - 4.f) + 0.5f) / (float)(wc * 8 - 4.5);
    float gy = ((ky - 4.f) + 0.5f) / (float)(hc * 8 - 4.5);
    gx = gx * 2.f - 1.f;
    gy = gy * 2.f - 1.f;
    const float ix = ((gx + 1.f) / 2.f) * (float)(wc - 1);
    const float iy = ((gy + 1.f) / 2.f) * (float)(hc - 1);
    const float xw = floorf(ix), yn = floorf(iy);
    const float w = ix - xw, e = 1.f - w, n = iy - yn, s = 1.f - n;
    const int x0 = (int)xw, y0 = (int)yn;
    const float wts[4] = {e * s, w * s, e * n, w * n};  // nw, ne, sw, se
    const float* base = dense + (long)b * hc * wc * 256;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
        if (xx < 0 || xx >= wc || yy < 0 || yy >= hc) continue;  // wave-uniform
        const float4 v = *reinterpret_cast<const float4*>(base + ((long)yy * wc + xx) * 256 + lane * 4);
        const float ss = wave_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
        const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
        const float f = wts[t];
        acc.x += (v.x * inv) * f; acc.y += (v.y * inv) * f; acc.z += (v.z * inv) * f; acc.w += (v.w * inv) * f;
    }
    const float ss = wave_sum(acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w);
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
    *reinterpret_cast<float4*>(desc + ((long)b * kmax + kp) * 256 + lane * 4) = acc;
}

hipError_t launch_sample_desc(const float* dense, int B, int hc, int wc, const float* kpts, const int* n_ptr, int kmax,
                              float* desc, hipStream_t st) {
    hipLaunchKernelGGL(sample_desc_kernel, dim3((kmax + 3) / 4, B), dim3(256), 0, st, dense, hc, wc, kpts, n_ptr, kmax, desc);
    return hipGetLastError();
}

}  // namespace im
