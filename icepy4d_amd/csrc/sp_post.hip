// SuperPoint post-processing: detector softmax + depth-to-space, simple_nms, border/threshold/top-k
// selection, descriptor sampling. All HBM-bound integer / compare work: coalesced loads, LDS tiles for the
// stencils, wavefront reductions; no matrix cores.
#include <cstdlib>

#include "common.h"
#include "kernels.h"
#include "sp_post.h"
#include "lg_misc.h"
#include "rank_sweep.h"

namespace im {

// ---------------------------------------------------------------------------------------------------------
// detector head tail (`lightglue/superpoint.py:170-173`): softmax over 65 logits per cell, drop the dustbin,
// channel c of cell (i, j) -> pixel (8i + c/8, 8j + c%8). HBM-bound: 65 floats in, 64 out per cell. A block takes 32 cells
// of one cell row = 2080 consecutive logits (coalesced 4-byte loads into LDS; rows of 65 floats are not 16-byte aligned) and writes
// 8 pixel rows of 256 pixels: thread (o, c) owns pixel row o of cell c, i.e. 8 channels in and two float4 (32 contiguous bytes) out,
// and the 32 lanes of a half wave write one contiguous KiB. The softmax statistics of a cell are combined over its 8 threads
// through LDS (partial max, then partial sum of exp); same expf / division per element as the one-wave-per-cell form of rounds
// 1-3 (18 us for the two 1080p maps of a pair, 35 MB: a quarter of the HBM rate), only the order of the additions differs.
static constexpr int DS_CELLS = 32;
__global__ __launch_bounds__(256) void det_softmax_shuffle_kernel(const float* __restrict__ logits, int ld,
                                                                   float* __restrict__ smap, int B, int hc, int wc, int groups_per_row) {
    __shared__ float sl[DS_CELLS * 65 + 3];
    __shared__ float sm[8][DS_CELLS + 1], ss[8][DS_CELLS + 1];
    const int tid = threadIdx.x;
    const int row = blockIdx.x / groups_per_row, g = blockIdx.x - row * groups_per_row;    // row = b * hc + i
    const int j0 = g * DS_CELLS, ncell = min(DS_CELLS, wc - j0);
    const float* lp = logits + ((long)row * wc + j0) * ld;
    if (ld == 65) {
        for (int idx = tid; idx < ncell * 65; idx += 256) sl[idx] = lp[idx];
    } else {
        for (int idx = tid; idx < ncell * 65; idx += 256) sl[idx] = lp[(long)(idx / 65) * ld + idx % 65];
    }
    __syncthreads();
    const int o = tid >> 5, c = tid & 31;
    const bool on = c < ncell;
    float v[8];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        v[k] = on ? sl[c * 65 + 8 * o + k] : 0.f;      // stride 65 across lanes: conflict-free
        mx = fmaxf(mx, v[k]);
    }
    sm[o][c] = mx;
    __syncthreads();
    const float dust = on ? sl[c * 65 + 64] : 0.f;
    float M = dust;
#pragma unroll
    for (int k = 0; k < 8; ++k) M = fmaxf(M, sm[k][c]);
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        v[k] = expf(v[k] - M);
        part += v[k];
    }
    ss[o][c] = part;
    __syncthreads();
    float sum = expf(dust - M);
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += ss[k][c];
    if (!on) return;
    const int W8 = wc * 8;
    float* dst = smap + ((long)row * 8 + o) * W8 + 8 * (j0 + c);
    reinterpret_cast<float4*>(dst)[0] = make_float4(v[0] / sum, v[1] / sum, v[2] / sum, v[3] / sum);
    reinterpret_cast<float4*>(dst)[1] = make_float4(v[4] / sum, v[5] / sum, v[6] / sum, v[7] / sum);
}

hipError_t launch_det_softmax(const float* logits, int ld, float* smap, int B, int hc, int wc, hipStream_t s) {
    const int groups = (wc + DS_CELLS - 1) / DS_CELLS;
    const long blocks = (long)B * hc * groups;
    if (blocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(det_softmax_shuffle_kernel, dim3((unsigned)blocks), dim3(256), 0, s, logits, ld, smap, B, hc, wc, groups);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// simple_nms (`lightglue/superpoint.py:50-65`), staged form: five (2r+1)^2 max-pools (stride 1, -inf padding) chained with
// exact fp32 equality tests, one launch per stage; the pooled quantity of a 32 x 64 tile plus halo r is staged in LDS and
// reduced separably. Only used for radius 5..8 and map widths that are not a multiple of 4; the forward pass with the
// reference's radii (3, 4) runs nms_round_kernel below.
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

static hipError_t launch_nms_staged(const float* s, float* out, uint8_t* mask, uint8_t* supp, float* rest, int B, int H, int W, int r,
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

struct SelState {            // per image, zeroed by launch_* before every use
    unsigned hist[4][256];   // radix histograms of the score bits, 8 bits per pass, most significant first
    int n_sel, n_eq, pad0, pad1;
};

__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// sliding maximum of width 2R+1 over v[0 .. N + 2R): out[i] = max(v[i .. i + 2R]), in place into v[0 .. N).
// Two v_max3 steps per output (R >= 2): w3[i] = max(v[i .. i + 2]), then three (two) w3 windows that tile the 2R+1 range.
template <int R, int N>
__device__ __forceinline__ void sliding_max(float (&v)[N + 8]) {
    static_assert(R >= 1 && R <= 4, "window 3 .. 9");
    constexpr int L = N + 2 * R;   // valid inputs
    if constexpr (R == 1) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = vmax3(v[i], v[i + 1], v[i + 2]);
    } else {
        // in place, ascending i reads only entries not yet overwritten
#pragma unroll
        for (int i = 0; i < L - 2; ++i) v[i] = vmax3(v[i], v[i + 1], v[i + 2]);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if constexpr (R == 2) v[i] = fmaxf(v[i], v[i + 2]);
            else v[i] = vmax3(v[i], v[i + R - 1], v[i + 2 * R - 2]);
        }
    }
}

__device__ __forceinline__ unsigned long long cand_key(float v, unsigned flat_idx) {
    return ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(0xFFFFFFFFu - flat_idx);
}

// ---------------------------------------------------------------------------------------------------------
// simple_nms as THREE launches, one per round (`lightglue/superpoint.py:50-65`), with the maxima kept as BIT masks in global memory
// between the rounds (1 bit per pixel: 0.26 MB per 1080p image):
//   round 0:  keep0 = (S == pool(S))
//   round 1:  near = dilate_r(keep0), rest = near ? 0 : S, keep1 = keep0 | (rest == pool(rest) & ~near)
//   round 2:  the same from keep1 -> keep2, out = keep2 ? S : 0 (+ the candidate epilogue)
// A round needs the scores on tile + r and the previous mask on tile + 2r, so a 32 x 64 tile stages 40 x 72 floats (1.4 x its own
// pixels) and runs ONE separable pool there - against the single-launch kernel of rounds 2-4 (tools/experiments/nms_fused_single_launch.hip.txt), whose 5r
// halo made it stage 72 x 96 floats (3.9 x) and run three pools on them behind twelve barriers: 40 against 26 us for two 1080p maps. Three barriers per block here, 23 KB of LDS (six blocks per CU), every
// thread busy in the compare pass. The score map is read three times, from L2 / Infinity Cache after the first. Exact by construction:
// max and == on fp32 only.
namespace nr {
constexpr int TH = 32, TW = 64, CH = 4, RW = TW + 2 * CH, PITCH = 76, NT = 256;
constexpr int SROWS = TH + 8, KROWS = TH + 16;     // at r = 4
}  // namespace nr

template <int R, int STAGE>
__global__ __launch_bounds__(nr::NT) void nms_round_kernel(const float* __restrict__ s, const unsigned* __restrict__ keep_in,
                                                           unsigned char* __restrict__ keep_out, float* __restrict__ out, int H, int W, int WW,
                                                           int tiles_x, int tiles_per_img, int total_tiles, int border, float thr,
                                                           unsigned long long* __restrict__ keys, long key_stride, int* __restrict__ n_cand,
                                                           SelState* __restrict__ sel, unsigned long long* __restrict__ gstage,
                                                           int* __restrict__ tile_cnt) {
    using namespace nr;
    constexpr int SR = TH + 2 * R;                  // rows of S: y0 - R .. y0 + TH + R - 1
    constexpr int KR = TH + 4 * R;                  // rows of the previous mask: y0 - 2R ..
    __shared__ __attribute__((aligned(16))) float lds[(SROWS + TH) * PITCH];
    __shared__ __attribute__((aligned(16))) unsigned kw[KROWS * 4];      // previous keep, words x0 / 32 - 1 .. + 2
    __shared__ __attribute__((aligned(16))) unsigned nw[SROWS * 4];      // near = dilate_r(previous keep), same words, rows of S
    float* S = lds;
    float* T = lds + SROWS * PITCH;
    const int tid = threadIdx.x;
    const int per_xcd = (int)gridDim.x / 8;          // XCD-aware block -> tile map: XCD x takes the x-th contiguous eighth of the tile list (a band of tile rows), so the halos
    // neighbouring tiles share are re-read from that XCD's L2 (blocks are dealt round-robin over the 8 XCDs; the grid is padded to a multiple of 8)
    const int tile_lin = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (tile_lin >= total_tiles) return;
    const int b = tile_lin / tiles_per_img;
    const int trem = tile_lin - b * tiles_per_img;
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
    const long img = (long)b * H * W;
    const float NINF = -INFINITY;

    // ---- scores of the region (-inf outside the image: max_pool2d's implicit padding) and the previous mask
    for (int idx = tid; idx < SR * (RW / 4); idx += NT) {
        const int ry = idx / (RW / 4), q = idx - ry * (RW / 4);
        const int gy = y0 - R + ry, gx = x0 - CH + 4 * q;
        float4 v = make_float4(NINF, NINF, NINF, NINF);
        if (gy >= 0 && gy < H && gx >= 0 && gx + 3 < W) v = *reinterpret_cast<const float4*>(s + img + (long)gy * W + gx);
        *reinterpret_cast<float4*>(S + ry * PITCH + 4 * q) = v;
    }
    if constexpr (STAGE > 0) {
        for (int idx = tid; idx < KR * 4; idx += NT) {
            const int kr = idx >> 2, w = idx & 3;
            const int gy = y0 - 2 * R + kr, gw = (x0 >> 5) - 1 + w;
            kw[idx] = (gy >= 0 && gy < H && gw >= 0 && gw < WW) ? keep_in[((long)b * H + gy) * WW + gw] : 0u;
        }
    }
    __syncthreads();

    if constexpr (STAGE > 0) {
        // ---- near on the rows of S: vertical OR of the 128-bit mask rows, then the horizontal dilation (the two commute); the scores under
        // it are zeroed in place (in-image pixels only: the padding stays -inf). One (row, word) per thread.
        if (tid < SR * 4) {
            const int ry = tid >> 2, w = tid & 3;
            unsigned k0 = 0, k1 = 0, k2 = 0, k3 = 0;
#pragma unroll
            for (int d = 0; d <= 2 * R; ++d) {
                const uint4 kr = *reinterpret_cast<const uint4*>(kw + (ry + d) * 4);
                k0 |= kr.x; k1 |= kr.y; k2 |= kr.z; k3 |= kr.w;
            }
            const unsigned lo = w == 0 ? 0u : (w == 1 ? k0 : (w == 2 ? k1 : k2));       // the word to the left / right of this one
            const unsigned me = w == 0 ? k0 : (w == 1 ? k1 : (w == 2 ? k2 : k3));
            const unsigned hi = w == 0 ? k1 : (w == 1 ? k2 : (w == 2 ? k3 : 0u));
            unsigned m = me;
#pragma unroll
            for (int i = 1; i <= R; ++i) m |= (me << i) | (me >> i) | (lo >> (32 - i)) | (hi << (32 - i));
            nw[tid] = m;
            const int gy = y0 - R + ry;
            const int g0 = x0 - 32 + 32 * w;                                            // global x of bit 0
            unsigned cm = 0xFFFFFFFFu;
            if (g0 < 0) cm = 0u;                                                        // (x0 is a multiple of 64: a word is inside or outside)
            if (g0 + 32 > W) cm = (g0 >= W) ? 0u : (0xFFFFFFFFu >> (g0 + 32 - W));
            const unsigned z = (gy >= 0 && gy < H) ? (m & cm) : 0u;
            if (z) {
                // word 1 / 2 = region columns 4 .. 35 / 36 .. 67; of word 0 only bits 28 .. 31 (columns 0 .. 3), of word 3 bits 0 .. 3 (68 .. 71)
                const int q_lo = w == 0 ? 7 : 0, q_hi = w == 3 ? 1 : 8;
                float4* row = reinterpret_cast<float4*>(S + ry * PITCH + CH + 32 * (w - 1));
                for (int q = q_lo; q < q_hi; ++q) {
                    const unsigned nib = (z >> (4 * q)) & 15u;
                    if (nib) {
                        float4 v = row[q];
                        if (nib & 1u) v.x = 0.f;
                        if (nib & 2u) v.y = 0.f;
                        if (nib & 4u) v.z = 0.f;
                        if (nib & 8u) v.w = 0.f;
                        row[q] = v;
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- vertical pass: T[t][c] = max over S rows t .. t + 2R (= tile row t +- R), every column of the region
    for (int t = tid; t < RW * 2; t += NT) {
        const int seg = t / RW, c = t - seg * RW;
        float v[16 + 8];
#pragma unroll
        for (int i = 0; i < 16 + 2 * R; ++i) v[i] = S[(16 * seg + i) * PITCH + c];
#pragma unroll
        for (int i = 16 + 2 * R; i < 24; ++i) v[i] = NINF;
        sliding_max<R, 16>(v);
#pragma unroll
        for (int i = 0; i < 16; ++i) T[(16 * seg + i) * PITCH + c] = v[i];
    }
    __syncthreads();

    // ---- horizontal pass + equality test: 8 outputs of one tile row per thread = one byte of the mask
    const int t_row = tid >> 3, seg = tid & 7;
    const int gy = y0 + t_row, gx = x0 + 8 * seg;
    unsigned kout;
    {
        const float4* tr = reinterpret_cast<const float4*>(T + t_row * PITCH + 8 * seg);      // region columns 8 seg .. 8 seg + 15
        float v[8 + 8];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float4 t4 = tr[q]; v[4 * q] = t4.x; v[4 * q + 1] = t4.y; v[4 * q + 2] = t4.z; v[4 * q + 3] = t4.w; }
        if constexpr (R < 4) {      // the window of output i starts at region column 8 seg + CH + i - R
#pragma unroll
            for (int i = 0; i < 8 + 2 * R; ++i) v[i] = v[i + CH - R];
        }
        sliding_max<R, 8>(v);
        const float4* cr = reinterpret_cast<const float4*>(S + (t_row + R) * PITCH + CH + 8 * seg);
        unsigned eq = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 c4 = cr[q];
            eq |= (c4.x == v[4 * q] ? 1u : 0u) << (4 * q);
            eq |= (c4.y == v[4 * q + 1] ? 1u : 0u) << (4 * q + 1);
            eq |= (c4.z == v[4 * q + 2] ? 1u : 0u) << (4 * q + 2);
            eq |= (c4.w == v[4 * q + 3] ? 1u : 0u) << (4 * q + 3);
        }
        unsigned vm = 0u;                                                             // in-image outputs of this byte
        if (gy < H && gx < W) vm = (gx + 8 <= W) ? 0xFFu : (0xFFu >> (gx + 8 - W));
        if constexpr (STAGE == 0) {
            kout = eq & vm;
        } else {
            const int w = 1 + (seg >> 2), sh = 8 * (seg & 3);
            const unsigned nearb = (nw[(t_row + R) * 4 + w] >> sh) & 0xFFu;
            const unsigned prev = (kw[(t_row + 2 * R) * 4 + w] >> sh) & 0xFFu;
            kout = prev | (eq & vm & ~nearb);
        }
    }
    if constexpr (STAGE < 2) {
        if (gy < H && gx < 32 * WW) keep_out[(((long)b * H + gy) * WW) * 4 + (gx >> 3)] = (unsigned char)kout;
        return;
    } else {
        // ---- epilogue: out = keep2 ? S : 0 (S from global: the LDS copy is zeroed around the maxima) and the candidate keys
        const bool e_on = gy < H && gx < W;              // W % 4 == 0: whole quads
        const bool e_on2 = e_on && gx + 4 < W;
        float4 sv0 = make_float4(0.f, 0.f, 0.f, 0.f), sv1 = sv0;
        if (e_on) sv0 = *reinterpret_cast<const float4*>(s + img + (long)gy * W + gx);
        if (e_on2) sv1 = *reinterpret_cast<const float4*>(s + img + (long)gy * W + gx + 4);
        const float ov[8] = {(kout & 1u) ? sv0.x : 0.f, (kout & 2u) ? sv0.y : 0.f, (kout & 4u) ? sv0.z : 0.f, (kout & 8u) ? sv0.w : 0.f,
                             (kout & 16u) ? sv1.x : 0.f, (kout & 32u) ? sv1.y : 0.f, (kout & 64u) ? sv1.z : 0.f, (kout & 128u) ? sv1.w : 0.f};
        int* cnt = reinterpret_cast<int*>(lds);                                   // [0] candidates of this block, [1] global base
        unsigned* lhist = reinterpret_cast<unsigned*>(lds) + 4;                   // [256]
        unsigned long long* stage = reinterpret_cast<unsigned long long*>(lds + 512);   // up to TH * TW keys: 16 KB of the 21.9
        static_assert(512 + 2 * TH * TW <= (SROWS + TH) * PITCH, "candidate staging fits in the S / T area");
        int n = 0;
        if (keys) {
            __syncthreads();                             // every thread is done with S and T
            for (int i = tid; i < 260; i += NT) reinterpret_cast<unsigned*>(lds)[i] = 0u;
            __syncthreads();
            if (kout && gy >= border && gy < H - border) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int x = gx + j;
                    if (ov[j] > thr && x >= border && x < W - border) {
                        const int p = atomicAdd(&cnt[0], 1);
                        stage[p] = cand_key(ov[j], (unsigned)(gy * W + x));
                        atomicAdd(&lhist[__float_as_uint(ov[j]) >> 24], 1u);
                    }
                }
            }
            __syncthreads();
            n = cnt[0];
            // where the block's keys go: into the tile's own 2048-entry slot of a staging buffer with its count next to it (cand_compact_kernel
            // makes the dense list) - or, without staging, behind the image's counter: ONE returning atomic per block on ONE address per
            // image, ~1000 of them queueing up in L2 while their blocks wait for the answer (~20 of the round's 32 us, round 5)
            if (!gstage && n > 0 && tid == 0) cnt[1] = atomicAdd(&n_cand[b], n);
            if (n > 0 && lhist[tid]) atomicAdd(&sel[b].hist[0][tid], lhist[tid]);
        }
        if (out && e_on) *reinterpret_cast<float4*>(out + img + (long)gy * W + gx) = make_float4(ov[0], ov[1], ov[2], ov[3]);
        if (out && e_on2) *reinterpret_cast<float4*>(out + img + (long)gy * W + gx + 4) = make_float4(ov[4], ov[5], ov[6], ov[7]);
        if (gstage) {
            if (tid == 0) tile_cnt[tile_lin] = n;
            unsigned long long* kd = gstage + (long)tile_lin * (TH * TW);
            for (int i = tid; i < n; i += NT) kd[i] = stage[i];
        } else if (n > 0) {
            __syncthreads();
            unsigned long long* kd = keys + (long)b * key_stride + cnt[1];
            for (int i = tid; i < n; i += NT) kd[i] = stage[i];
        }
    }
}

// candidates of an existing NMS map (stage entry point im_select_topk; also the path for radius > 4)
__global__ __launch_bounds__(256) void kp_extract_kernel(const float* __restrict__ nms, int H, int W, int border, float thr,
                                                          unsigned long long* __restrict__ keys, long key_stride,
                                                          int* __restrict__ n_cand, SelState* __restrict__ sel) {
    __shared__ int cnt[2];
    __shared__ unsigned lhist[256];
    __shared__ unsigned long long stage[1024];
    const int b = blockIdx.y, tid = threadIdx.x;
    const long npix = (long)H * W;
    const float* s = nms + (long)b * npix;
    if (tid < 2) cnt[tid] = 0;
    lhist[tid] = 0u;
    __syncthreads();
    const long base = (long)blockIdx.x * 1024 + tid * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long idx = base + j;
        if (idx >= npix) break;
        const float v = s[idx];
        const int y = (int)(idx / W), x = (int)(idx - (long)y * W);
        if (v > thr && y >= border && y < H - border && x >= border && x < W - border) {
            const int p = atomicAdd(&cnt[0], 1);
            stage[p] = cand_key(v, (unsigned)idx);
            atomicAdd(&lhist[__float_as_uint(v) >> 24], 1u);
        }
    }
    __syncthreads();
    const int n = cnt[0];
    if (n == 0) return;
    if (tid == 0) cnt[1] = atomicAdd(&n_cand[b], n);
    if (lhist[tid]) atomicAdd(&sel[b].hist[0][tid], lhist[tid]);
    __syncthreads();
    unsigned long long* kd = keys + (long)b * key_stride + cnt[1];
    for (int i = tid; i < n; i += 256) kd[i] = stage[i];
}

// ---------------------------------------------------------------------------------------------------------
// top-k selection (`top_k_keypoints`, `lightglue/superpoint.py:68-72`; SuperGlue flavour `models/superpoint.py:196-203`)
// over the UNORDERED candidate keys of an image. Key = score bits (positive floats order like unsigned integers) in the
// high word, ~flat pixel index in the low word: unique; among equal scores the lower index wins (torch.topk's order among
// ties is unspecified).
//   n <= k : every candidate, in row-major order (torch.where / nonzero order): ranked by pixel index
//   n >  k : multi-block radix select of the k-th largest score (4 passes of 8 bits; pass 0 is counted where the candidates
//            are produced), then the keys above the cut are collected and ranked by counting (rank = number of larger keys).
//            Only if the k-th score is shared by more candidates than fit, those equal ones are ranked by index among
//            themselves and the first ones fill the remaining slots.
// Every kernel finds the cut again from the histograms (one wave, 256 bins per pass): no host round trip, no extra launch.
struct Cut { unsigned prefix; int remaining; int in_bin; };

// all 64 lanes of one wave; digits of passes 0 .. passes-1 -> prefix (score bits fixed so far), how many keys of the cut
// bin are still wanted, how many keys that bin holds
__device__ __forceinline__ Cut find_cut(const SelState* __restrict__ st, int k, int passes) {
    const int lane = threadIdx.x & 63;
    Cut c{0u, k, 0};
    for (int p = 0; p < passes; ++p) {
        const uint4 h = *reinterpret_cast<const uint4*>(&st->hist[p][lane * 4]);
        const int sum = (int)(h.x + h.y + h.z + h.w);
        int inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        const int total = __shfl(inc, 63);
        const int above = total - inc;                                    // keys in bins of higher lanes
        const bool mine = above < c.remaining && c.remaining <= above + sum;
        const unsigned long long bal = __ballot(mine);
        const int src = bal ? (int)__builtin_ctzll(bal) : 0;              // (remaining <= total always holds when n > k)
        int digit = 0, rem = c.remaining, inb = 0;
        if (mine) {
            const unsigned hb[4] = {h.x, h.y, h.z, h.w};
            int cum = above;
#pragma unroll
            for (int j = 3; j >= 0; --j) {
                if (inb == 0 && cum + (int)hb[j] >= c.remaining && hb[j] > 0) { digit = lane * 4 + j; rem = c.remaining - cum; inb = (int)hb[j]; }
                if (inb == 0) cum += (int)hb[j];
            }
        }
        digit = __shfl(digit, src); rem = __shfl(rem, src); inb = __shfl(inb, src);
        c.prefix |= (unsigned)digit << (24 - 8 * p);
        c.remaining = rem;
        c.in_bin = inb;
    }
    return c;
}

static constexpr int SEL_BLOCKS = 64;     // blocks per image in the histogram / collect / tie kernels (grid-stride over the keys)

__global__ __launch_bounds__(256) void sel_hist_kernel(const unsigned long long* __restrict__ keys, long key_stride,
                                                        const int* __restrict__ n_cand, SelState* __restrict__ sel, int k_req, int kmax,
                                                        int pass) {
    __shared__ unsigned lhist[256];
    __shared__ unsigned sh_prefix;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int n = n_cand[b];
    const int k = (k_req > 0 && k_req < kmax) ? k_req : kmax;
    if (n <= k) return;
    lhist[tid] = 0u;
    if (tid < 64) {
        const Cut c = find_cut(&sel[b], k, pass);
        if (tid == 0) sh_prefix = c.prefix;
    }
    __syncthreads();
    const int shift = 24 - 8 * pass;
    const unsigned want = sh_prefix >> (shift + 8);
    const unsigned long long* kd = keys + (long)b * key_stride;
    for (int i = blockIdx.x * 256 + tid; i < n; i += gridDim.x * 256) {
        const unsigned sc = (unsigned)(kd[i] >> 32);
        if ((sc >> (shift + 8)) == want) atomicAdd(&lhist[(sc >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (lhist[tid]) atomicAdd(&sel[b].hist[pass][tid], lhist[tid]);
}

// keys above the cut -> selected list; keys AT the cut score go there too when all of them fit, else to the tie list
__global__ __launch_bounds__(256) void sel_collect_kernel(const unsigned long long* __restrict__ keys, long key_stride,
                                                           const int* __restrict__ n_cand, SelState* __restrict__ sel, int k_req, int kmax,
                                                           unsigned long long* __restrict__ chosen, long chosen_stride,
                                                           unsigned long long* __restrict__ ties, long ties_stride) {
    __shared__ Cut sh_cut;
    __shared__ int cnt[4];
    __shared__ unsigned long long st_sel[1024], st_eq[1024];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int n = n_cand[b];
    const int k = (k_req > 0 && k_req < kmax) ? k_req : kmax;
    if (n <= k) return;
    if (tid < 64) {
        const Cut c = find_cut(&sel[b], k, 4);
        if (tid == 0) sh_cut = c;
    }
    const unsigned long long* kd = keys + (long)b * key_stride;
    // chunks of 1024 keys so that the LDS staging lists cannot overflow
    for (int base = blockIdx.x * 1024; base < n; base += gridDim.x * 1024) {
        __syncthreads();
        if (tid < 4) cnt[tid] = 0;
        __syncthreads();
        const unsigned cut = sh_cut.prefix;
        const bool all_equal_fit = sh_cut.remaining == sh_cut.in_bin;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = base + j * 256 + tid;
            if (i < n) {
                const unsigned long long key = kd[i];
                const unsigned sc = (unsigned)(key >> 32);
                if (sc > cut || (sc == cut && all_equal_fit)) st_sel[atomicAdd(&cnt[0], 1)] = key;
                else if (sc == cut) st_eq[atomicAdd(&cnt[1], 1)] = key;
            }
        }
        __syncthreads();
        if (tid == 0) {
            cnt[2] = cnt[0] ? atomicAdd(&sel[b].n_sel, cnt[0]) : 0;
            cnt[3] = cnt[1] ? atomicAdd(&sel[b].n_eq, cnt[1]) : 0;
        }
        __syncthreads();
        for (int i = tid; i < cnt[0]; i += 256) chosen[(long)b * chosen_stride + cnt[2] + i] = st_sel[i];
        for (int i = tid; i < cnt[1]; i += 256) ties[(long)b * ties_stride + cnt[3] + i] = st_eq[i];
    }
}

__device__ __forceinline__ void emit_kp(unsigned long long key, int W, float* kp, float* sc) {
    const unsigned idx = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
    const int y = idx / W, x = idx - y * W;
    kp[0] = (float)x;
    kp[1] = (float)y;
    *sc = __uint_as_float((unsigned)(key >> 32));
}

// ties at the cut: the `remaining` equal-score keys with the lowest pixel index fill slots k - remaining .. k - 1
__global__ __launch_bounds__(RK_N) void sel_tie_kernel(const int* __restrict__ n_cand, const SelState* __restrict__ sel, int k_req, int kmax,
                                                       const unsigned long long* __restrict__ ties, long ties_stride, int W,
                                                       float* __restrict__ kpts, float* __restrict__ scores) {
    __shared__ __attribute__((aligned(16))) unsigned long long tile[1024];
    __shared__ int part[RK_N];
    __shared__ Cut sh_cut;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int n = n_cand[b];
    const int k = (k_req > 0 && k_req < kmax) ? k_req : kmax;
    if (n <= k) return;
    if (tid < 64) {
        const Cut c = find_cut(&sel[b], k, 4);
        if (tid == 0) sh_cut = c;
    }
    __syncthreads();
    const int r = sh_cut.remaining, m = sh_cut.in_bin;
    if (r == m) return;                                           // every key of the cut score was taken by sel_collect
    const unsigned long long* src = ties + (long)b * ties_stride;
    for (int base = blockIdx.x * RK_T; base < m; base += gridDim.x * RK_T) {   // block-uniform trip count (rank_among synchronises)
        const int i = base + (tid & 63);
        const unsigned long long mine = i < m ? src[i] : 0ull;
        const int rank = rank_among(src, m, mine, ~0ull, tile, part);
        if (tid < RK_T && i < m && rank < r) {
            const int pos = k - r + rank;
            emit_kp(mine, W, kpts + ((long)b * kmax + pos) * 2, scores + (long)b * kmax + pos);
        }
    }
}

// final order: n <= k -> all candidates by ascending pixel index; else the selected keys by descending key
__global__ __launch_bounds__(RK_N) void sel_rank_kernel(const unsigned long long* __restrict__ keys, long key_stride,
                                                        const int* __restrict__ n_cand, const SelState* __restrict__ sel, int k_req, int kmax,
                                                        const unsigned long long* __restrict__ chosen, long chosen_stride, int W,
                                                        float* __restrict__ kpts, float* __restrict__ scores, int* __restrict__ n_out) {
    __shared__ __attribute__((aligned(16))) unsigned long long tile[1024];
    __shared__ int part[RK_N];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int n = n_cand[b];
    const int k = (k_req > 0 && k_req < kmax) ? k_req : kmax;
    const bool all = n <= k;
    const unsigned long long* src = all ? keys + (long)b * key_stride : chosen + (long)b * chosen_stride;
    const int m = all ? n : sel[b].n_sel;
    const unsigned long long mask = all ? 0xFFFFFFFFull : ~0ull;   // low word = ~index: larger = earlier in row-major order
    if (blockIdx.x == 0 && tid == 0) n_out[b] = all ? n : k;
    if (blockIdx.x * RK_T >= m) return;
    const int i = blockIdx.x * RK_T + (tid & 63);
    const unsigned long long mine = i < m ? src[i] : 0ull;
    const int rank = rank_among(src, m, mine, mask, tile, part);
    if (tid < RK_T && i < m) emit_kp(mine, W, kpts + ((long)b * kmax + rank) * 2, scores + (long)b * kmax + rank);
}

// per-device opt-in for more than 64 KB of dynamic LDS (a process may hold contexts on several GPUs)
hipError_t ensure_dyn_lds(const void* fn, size_t bytes, size_t* cache) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= IM_MAX_DEVICES) return hipErrorInvalidDevice;
    if (bytes > cache[dev]) {
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        cache[dev] = bytes;
    }
    return hipSuccess;
}

size_t sel_state_bytes(int B) { return (size_t)B * sizeof(int) + 16 + (size_t)B * sizeof(SelState); }

static hipError_t sel_reset(const SelBuffers& sb, int B, hipStream_t st) {   // a kernel, not a memset node (see lg_misc.hip)
    return launch_zero_words(sb.n_cand, (long)(sel_state_bytes(B) / 4), st);
}

static SelState* sel_states(int* n_cand, int B) {
    return reinterpret_cast<SelState*>(reinterpret_cast<char*>(n_cand) + (((size_t)B * sizeof(int) + 15) & ~(size_t)15));
}

// The dense candidate list of an image from the per-tile slots the last NMS round filled: block (t, b) sums the counts of the tiles in front
// of its own (<= 16 per lane) and copies its keys behind them; the last tile's block writes the image's count. Tile order instead of arrival
// order - every consumer of the list ranks or reduces with order-independent integer operations.
__global__ __launch_bounds__(64) void cand_compact_kernel(const unsigned long long* __restrict__ gstage, const int* __restrict__ tile_cnt,
                                                         int tiles_per_img, unsigned long long* __restrict__ keys, long key_stride,
                                                         int* __restrict__ n_cand) {
    const int t = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int* tc = tile_cnt + (long)b * tiles_per_img;
    int part = 0;
    for (int i = lane; i < t; i += 64) part += tc[i];
    const int prefix = wave_sum_i(part);
    const int n = tc[t];
    const unsigned long long* src = gstage + ((long)b * tiles_per_img + t) * (nr::TH * nr::TW);
    unsigned long long* dst = keys + (long)b * key_stride + prefix;
    for (int j = lane; j < n; j += 64) dst[j] = src[j];
    if (t == tiles_per_img - 1 && lane == 0) n_cand[b] = prefix + n;
}

template <int R>
static hipError_t launch_nms_rounds_r(const float* s, float* out, uint8_t* keep_a, uint8_t* keep_b, size_t mask_bytes, int B, int H, int W, int border,
                                      float thr, unsigned long long* keys, long key_stride, int* n_cand, unsigned long long* gstage, size_t gstage_cap,
                                      hipStream_t st) {
    const int tx = (W + nr::TW - 1) / nr::TW, ty = (H + nr::TH - 1) / nr::TH, WW = (W + 31) / 32;
    const int total = tx * ty * B;
    const dim3 grid(((total + 7) / 8) * 8), block(nr::NT);
    unsigned* ka = reinterpret_cast<unsigned*>(keep_a);      // [B][H][WW] words in the byte-mask buffers of the staged form (4 WW <= W)
    unsigned* kb = reinterpret_cast<unsigned*>(keep_b);
    SelState* ss = keys ? sel_states(n_cand, B) : nullptr;
    // per-tile candidate staging (no returning atomic in the last round) when the staging buffer holds a slot per tile and the tile counts fit
    // behind the bit masks; otherwise the blocks append behind the image's counter
    const size_t bits = (((size_t)B * H * WW * 4) + 15) & ~(size_t)15;
    const bool staged = keys && gstage && (size_t)total * (nr::TH * nr::TW) <= gstage_cap && bits + (size_t)total * sizeof(int) <= mask_bytes;
    int* tile_cnt = staged ? reinterpret_cast<int*>(keep_a + bits) : nullptr;      // (keep_a is read by round 1 only: free again in round 2)
    hipLaunchKernelGGL((nms_round_kernel<R, 0>), grid, block, 0, st, s, (const unsigned*)nullptr, keep_a, (float*)nullptr, H, W, WW, tx, tx * ty, total, border, thr,
                       (unsigned long long*)nullptr, 0L, (int*)nullptr, (SelState*)nullptr, (unsigned long long*)nullptr, (int*)nullptr);
    hipLaunchKernelGGL((nms_round_kernel<R, 1>), grid, block, 0, st, s, (const unsigned*)ka, keep_b, (float*)nullptr, H, W, WW, tx, tx * ty, total, border, thr,
                       (unsigned long long*)nullptr, 0L, (int*)nullptr, (SelState*)nullptr, (unsigned long long*)nullptr, (int*)nullptr);
    hipLaunchKernelGGL((nms_round_kernel<R, 2>), grid, block, 0, st, s, (const unsigned*)kb, (unsigned char*)nullptr, out, H, W, WW, tx, tx * ty, total, border, thr,
                       keys, key_stride, n_cand, ss, staged ? gstage : (unsigned long long*)nullptr, tile_cnt);
    if (staged) hipLaunchKernelGGL(cand_compact_kernel, dim3(tx * ty, B), dim3(64), 0, st, gstage, tile_cnt, tx * ty, keys, key_stride, n_cand);
    return hipGetLastError();
}

static hipError_t launch_nms_rounds(const float* s, float* out, uint8_t* keep_a, uint8_t* keep_b, size_t mask_bytes, int B, int H, int W, int r, int border,
                                    float thr, unsigned long long* keys, long key_stride, int* n_cand, unsigned long long* gstage, size_t gstage_cap,
                                    hipStream_t st) {
    switch (r) {
        case 1: return launch_nms_rounds_r<1>(s, out, keep_a, keep_b, mask_bytes, B, H, W, border, thr, keys, key_stride, n_cand, gstage, gstage_cap, st);
        case 2: return launch_nms_rounds_r<2>(s, out, keep_a, keep_b, mask_bytes, B, H, W, border, thr, keys, key_stride, n_cand, gstage, gstage_cap, st);
        case 3: return launch_nms_rounds_r<3>(s, out, keep_a, keep_b, mask_bytes, B, H, W, border, thr, keys, key_stride, n_cand, gstage, gstage_cap, st);
        default: return launch_nms_rounds_r<4>(s, out, keep_a, keep_b, mask_bytes, B, H, W, border, thr, keys, key_stride, n_cand, gstage, gstage_cap, st);
    }
}

static bool nms_fusable(int H, int W, int r) {
    static const bool staged = getenv("IM_NMS_STAGED") && getenv("IM_NMS_STAGED")[0] == '1';   // A/B switch: the five-launch form
    return !staged && r >= 1 && r <= 4 && (W % 4) == 0;
}

hipError_t launch_nms(const float* s, float* out, uint8_t* mask, uint8_t* supp, float* rest, int B, int H, int W, int r,
                      hipStream_t st) {
    if (r < 0 || r > NMS_RMAX) return hipErrorInvalidValue;
    if (nms_fusable(H, W, r)) return launch_nms_rounds(s, out, mask, supp, (size_t)B * H * W, B, H, W, r, 0, 0.f, nullptr, 0, nullptr, nullptr, 0, st);
    return launch_nms_staged(s, out, mask, supp, rest, B, H, W, r, st);
}

// radix passes 1..3, collect, ties, rank on the candidate keys (pass 0 was counted by the producer of the keys)
static hipError_t launch_select(int B, int H, int W, int k_req, int kmax, const SelBuffers& sb, float* kpts, float* scores, int* n_out,
                                hipStream_t st) {
    const long npix = (long)H * W;
    SelState* ss = sel_states(sb.n_cand, B);
    for (int pass = 1; pass < 4; ++pass)
        hipLaunchKernelGGL(sel_hist_kernel, dim3(SEL_BLOCKS, B), dim3(256), 0, st, sb.keys, npix, sb.n_cand, ss, k_req, kmax, pass);
    hipLaunchKernelGGL(sel_collect_kernel, dim3(SEL_BLOCKS, B), dim3(256), 0, st, sb.keys, npix, sb.n_cand, ss, k_req, kmax, sb.chosen,
                       (long)kmax, sb.ties, npix);
    hipLaunchKernelGGL(sel_tie_kernel, dim3(4 * SEL_BLOCKS, B), dim3(RK_N), 0, st, sb.n_cand, ss, k_req, kmax, sb.ties, npix, W, kpts, scores);
    hipLaunchKernelGGL(sel_rank_kernel, dim3((kmax + RK_T - 1) / RK_T, B), dim3(RK_N), 0, st, sb.keys, npix, sb.n_cand, ss, k_req, kmax, sb.chosen,
                       (long)kmax, W, kpts, scores, n_out);
    return hipGetLastError();
}

// the forward pass: score map -> NMS map (+ candidates) -> keypoints, 7 launches
hipError_t launch_nms_select(const float* s, float* nms_out, uint8_t* mask, uint8_t* supp, float* rest, int B, int H, int W, int r,
                             int border, float thr, int k_req, int kmax, const SelBuffers& sb, float* kpts, float* scores, int* n_out,
                             hipStream_t st) {
    if (r < 0 || r > NMS_RMAX) return hipErrorInvalidValue;
    hipError_t e = sel_reset(sb, B, st);
    if (e != hipSuccess) return e;
    const long npix = (long)H * W;
    if (nms_fusable(H, W, r)) {
        e = launch_nms_rounds(s, nms_out, mask, supp, (size_t)B * H * W, B, H, W, r, border, thr, sb.keys, npix, sb.n_cand, sb.ties, sb.ties_cap, st);
        if (e != hipSuccess) return e;
    } else {
        e = launch_nms_staged(s, nms_out, mask, supp, rest, B, H, W, r, st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kp_extract_kernel, dim3((unsigned)((npix + 1023) / 1024), B), dim3(256), 0, st, nms_out, H, W, border, thr, sb.keys,
                           npix, sb.n_cand, sel_states(sb.n_cand, B));
    }
    return launch_select(B, H, W, k_req, kmax, sb, kpts, scores, n_out, st);
}

hipError_t launch_select_topk(const float* nms, int B, int H, int W, int border, float thr, int k_req, int kmax, const SelBuffers& sb,
                              float* kpts, float* scores, int* n_out, hipStream_t st) {
    hipError_t e = sel_reset(sb, B, st);
    if (e != hipSuccess) return e;
    const long npix = (long)H * W;
    hipLaunchKernelGGL(kp_extract_kernel, dim3((unsigned)((npix + 1023) / 1024), B), dim3(256), 0, st, nms, H, W, border, thr, sb.keys, npix,
                       sb.n_cand, sel_states(sb.n_cand, B));
    return launch_select(B, H, W, k_req, kmax, sb, kpts, scores, n_out, st);
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
