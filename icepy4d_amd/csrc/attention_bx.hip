// Flash attention with fp32 accuracy on the bf16 matrix cores (round 5). Same operation, arguments, block map, split-KV protocol
// and output layout as attention.hip (the reference materialises the attention matrix on the CPU: `lightglue/lightglue.py:120-123`
// via SDPA, `:200-210` explicit einsum / softmax, `SuperGlue/models/superglue.py:87-93`).
//
// Why: gfx950's f32-input MFMA runs at the VECTOR rate (157 TFLOP/s, 1 / 16 of the bf16 matrix cores) and holds the vector issue port
// while it runs; the bf16 MFMA does neither. An fp32 value is the exact sum of three bf16 values (8 + 8 + 8 significant bits, each cut
// rounded to nearest: x = x0 + x1 + x2 with |x1| <= 2^-8 |x|, |x2| <= 2^-17 |x|, nothing left), a bf16 product is exact in
// the matrix core's fp32 accumulation, and of the nine products of two such triples the three that are dropped (i + j >= 3) are worth at
// most 2^-24 |a b| - the rounding of ONE fp32 operation -, 2^-27.4 in the root mean square (tests/test_host_cpu.py pins this arithmetic
// in numpy). `tools/bf16x_probe.hip` measured the accumulation on the part
// (`profiles/r05_bf16x_probe.txt`): error against an f64 sum, in units of 2^-24 sum |a b|, rms 0.37-0.39 for six products against
// 0.45-0.47 for the f32 MFMA chain at K = 64 .. 4096 (the matrix core adds 16 products before it rounds once), nine products no
// better than six, three 4-30 x worse; and six `v_mfma_f32_32x32x16_bf16` per 16 k run at 2.2-2.35 PFLOP/s = 2.4-2.5 x the f32 MFMA,
// 1.8-2.15 x with four to five vector instructions between the MFMAs - the vector instructions of the splits and of the softmax run
// BESIDE the matrix cores instead of in front of them.
//
// Both products keep the QUERY on the MFMA lane, as attention.hip does:
//     S^T (32 keys x 32 queries) = K . Q^T     A = three bf16 planes of a K tile in LDS (16-byte row reads), B = Q planes in registers
//     O^T (64 d x 32 queries)   += V^T . P^T   A = three bf16 planes of a V tile in LDS, read TRANSPOSED by ds_read_b64_tr_b16,
//                                              B = the planes of P = exp2(S^T - m), cut from the accumulator registers in place
// (an accumulator tile is the next MFMA's B operand with the k order `16 s + 8 (j >> 2) + 4 h + (j & 3)`; the V reads use the same order).
// A block is 128 queries of one (image, head): 4 waves x 32 queries, 32 keys per step, two blocks per CU, each with its own barrier, so that one
// block's vector phases sit beside the other's MFMAs. Three forms of the K / V staging, same products in the same order (bit-identical results):
//   <PRE, DMA>   the product: K / V arrive as bf16 planes (kv_planes_kernel, one small launch before this one) and go into LDS by LDS-DMA - no
//                staging registers, no ds_write; unpadded 128-byte rows whose 16-byte pieces are permuted by the transfer, rings of three stages;
//   <PRE, !DMA>  the same planes through registers (IM_ATTN_REG_STAGING=1, A/B): padded rows, rings of two stages;
//   <!PRE>       no plane workspace: the fp32 tiles are cut while they are staged, once per block.
#include "common.h"
#include "kernels.h"
#include "sp_post.h"

#include <cstdlib>

namespace im {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

static constexpr int BKT = 32;                    // keys per step
static constexpr int BKS = 144;                   // bytes per key row of a K plane: 64 bf16 + 16 (conflict-free 16-byte row reads)
static constexpr int BVS = 192;                   // bytes per key row of a V plane: four consecutive rows fall on disjoint bank windows
static constexpr int BK_PLANE = BKT * BKS, BV_PLANE = BKT * BVS;
static constexpr int B_STAGE = 3 * BK_PLANE + 3 * BV_PLANE;     // 32,256 bytes
static constexpr int B_LDS = 2 * B_STAGE;
// DMA form: unpadded 128-byte rows (a plane of a tile = 4 KB = four 1 KB transfers that land lane-linear), the 16-byte pieces of a row permuted
// instead (the transfer fetches piece `pc ^ swz(row)` into place pc: swz = (row >> 1) & 7 for K - conflict-free 16-byte row reads of 16 rows -,
// ((row >> 1) & 1) << 2 for V - the four rows of a transposed read fall on disjoint bank windows), rings of THREE stages (72 KB: two blocks per CU)
static constexpr int DPL = BKT * 128;                           // bytes per plane of a tile
static constexpr int D_LDS = 3 * 3 * (DPL + DPL);

typedef unsigned int du32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* dlds_ptr_t;
__device__ __forceinline__ du32x4 dma_rsrc(const void* base, unsigned bytes) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    du32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) & 0xFFFFu;
    r.z = __builtin_amdgcn_readfirstlane(bytes);
    r.w = 0x00020000u;
    return r;
}
// one LDS-DMA piece as inline asm (conv_wino.hip: the compiler's wait-count pass must not see it, or it orders every later LDS read behind it)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void bx_dma16(du32x4 rsrc, unsigned lds_byte_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_byte_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
#pragma clang diagnostic pop

__device__ __forceinline__ f32x16 mfma_bf(u32x4 a, u32x4 b, f32x16 c) {
#ifdef BX_ABL_NO_MFMA       // timing ablation (wrong results): one vector instruction in place of every MFMA
    c[0] += __uint_as_float(a.x ^ b.x);
    return c;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}
__device__ __forceinline__ unsigned cvt_pk(float a, float b) {            // v_cvt_pk_bf16_f32: a in the low half, round to nearest even
    const bf16x2 v = __builtin_convertvector(f32x2{a, b}, bf16x2);
    return __builtin_bit_cast(unsigned, v);
}
// (a, b) -> three packed bf16 pairs with a = h.lo + m.lo + l.lo exactly: the subtractions are exact in fp32
__device__ __forceinline__ void split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pk(a, b);
    float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk(ra, rb);
    ra -= __uint_as_float(m << 16);
    rb -= __uint_as_float(m & 0xffff0000u);
    l = cvt_pk(ra, rb);
}

__device__ __forceinline__ float4 bx_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bx_rsrc_bytes(const void* base, int bytes) {     // reads past `bytes` return zero
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    void* p = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

struct Planes { u32x4 h, m, l; };
__device__ __forceinline__ Planes split8(float x0, float x1, float x2, float x3, float x4, float x5, float x6, float x7) {
    unsigned h[4], m[4], l[4];
    split2(x0, x1, h[0], m[0], l[0]);
    split2(x2, x3, h[1], m[1], l[1]);
    split2(x4, x5, h[2], m[2], l[2]);
    split2(x6, x7, h[3], m[3], l[3]);
    return Planes{u32x4{h[0], h[1], h[2], h[3]}, u32x4{m[0], m[1], m[2], m[3]}, u32x4{l[0], l[1], l[2], l[3]}};
}

// one float4 of a tile row -> 8 bytes in each of the three planes
__device__ __forceinline__ void stage4(unsigned char* plane0, int plane_bytes, int off, float4 x) {
    unsigned h0, m0, l0, h1, m1, l1;
    split2(x.x, x.y, h0, m0, l0);
    split2(x.z, x.w, h1, m1, l1);
    *reinterpret_cast<u32x2*>(plane0 + off) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(plane0 + plane_bytes + off) = u32x2{m0, m1};
    *reinterpret_cast<u32x2*>(plane0 + 2 * plane_bytes + off) = u32x2{l0, l1};
}

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
__device__ __forceinline__ u32x2 tr_read(const unsigned char* p) {
#ifdef BX_ABL_NO_VREAD      // timing ablation (wrong results): no transposed V reads
    const unsigned x = (unsigned)(unsigned long long)p;
    return u32x2{x, 0x3f803f80u};
#else
    const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p));
    return __builtin_bit_cast(u32x2, v);
#endif
}
__device__ __forceinline__ u32x4 kf_read(const unsigned char* p) {
#ifdef BX_ABL_NO_KREAD      // timing ablation (wrong results): no K fragment reads
    const unsigned x = (unsigned)(unsigned long long)p;
    return u32x4{x, 0x3f803f80u, x, 0x3f803f80u};
#else
    return *reinterpret_cast<const u32x4*>(p);
#endif
}

static constexpr float BX_SLACK = 8.f;   // as attention.hip: the running max is a reference, raised when a row exceeds it by 2^8
__device__ __forceinline__ float bmax3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// softmax of one 32-key tile, first half: mask the keys past the end (last step only), row maximum, and - rarely - a new reference
template <bool TAIL>
__device__ __forceinline__ float softmax_ref(f32x16& s, int kb, int nk, int hh, float& m_run, float& l_run, f32x16& o0, f32x16& o1) {
    if (TAIL) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (kb + acc_row(r, hh) >= nk) s[r] = -INFINITY;
    }
    float mx0 = bmax3(s[0], s[1], s[2]), mx1 = bmax3(s[3], s[4], s[5]);
    mx0 = bmax3(mx0, s[6], s[7]); mx1 = bmax3(mx1, s[8], s[9]);
    mx0 = bmax3(mx0, s[10], s[11]); mx1 = bmax3(mx1, s[12], s[13]);
    float mx = bmax3(mx0, mx1, fmaxf(s[14], s[15]));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (__builtin_amdgcn_ballot_w64(mx > m_run + BX_SLACK) != 0) {     // wave-uniform: raise the reference, rescale
        asm volatile("" ::: "memory");
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);     // m_run = -inf at the first tile: alpha = 0
        l_run *= alpha;
        o0 *= alpha;
        o1 *= alpha;
        m_run = m_new;
    }
    return (TAIL && m_run == -INFINITY) ? 0.f : m_run;                 // a split whose every key lies past the end keeps m = -inf, l = 0
}
// second half: P = exp2(S - m) in place, row sum
__device__ __forceinline__ void softmax_exp(f32x16& s, float m_use, float& l_run) {
    float r0 = 0.f, r1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        s[r] = __builtin_amdgcn_exp2f(s[r] - m_use);
        s[r + 1] = __builtin_amdgcn_exp2f(s[r + 1] - m_use);
        r0 += s[r]; r1 += s[r + 1];
    }
    float rs = r0 + r1;
    rs += __shfl_xor(rs, 32);
    l_run += rs;
}

// K and V of every (image, head) as bf16 triples, once per launch instead of once per 128-query block: [image][head][row][plane][64].
// blockIdx.y: 0 = K, 1 = V; eight threads per row. Rows past an image's live count are neither read nor written.
__global__ __launch_bounds__(256) void kv_planes_kernel(AttnArgs a) {
    const long rid = (long)blockIdx.x * 32 + (threadIdx.x >> 3);            // (z * heads + head) * n_max + row
    const int d8 = threadIdx.x & 7;
    const long per_image = (long)a.heads * a.n_max;
    if (rid >= per_image * a.batch) return;
    const int z = (int)(rid / per_image), rem = (int)(rid - (long)z * per_image), head = rem / a.n_max, row = rem - head * a.n_max;
    if (a.active && a.active[(z >> 1) * a.pstride] == 0) return;
    const int n = a.n_ptr ? a.n_ptr[(z >> 1) * a.pstride + (z & 1)] : a.n_max;
    if (row >= n) return;
    const float* src = (blockIdx.y ? a.v : a.k) + (long)z * a.bstride + (long)head * a.hstride + (long)row * 64 + d8 * 8;
    const float4 u = *reinterpret_cast<const float4*>(src), w = *reinterpret_cast<const float4*>(src + 4);
    const Planes p = split8(u.x, u.y, u.z, u.w, w.x, w.y, w.z, w.w);
    unsigned char* dst = reinterpret_cast<unsigned char*>(a.planes) + ((long)blockIdx.y * per_image * a.batch + rid) * 384 + d8 * 16;
    *reinterpret_cast<u32x4*>(dst) = p.h;
    *reinterpret_cast<u32x4*>(dst + 128) = p.m;
    *reinterpret_cast<u32x4*>(dst + 256) = p.l;
}

// PRE: K / V arrive as planes (kv_planes_kernel; staging is a copy). !PRE: fp32 K / V cut while they are staged (no workspace).
template <bool PRE, bool DMA>
__global__ __launch_bounds__(256, 2) void flash_attn_bx_kernel(AttnArgs a) {
    static_assert(PRE || !DMA, "the DMA form stages planes");
    extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
    constexpr int RD = DMA ? 3 : 2;                                  // ring depth
    constexpr int KS = DMA ? 128 : BKS, VS = DMA ? 128 : BVS;        // row strides
    constexpr int KPL = BKT * KS, VPL = BKT * VS;                    // plane sizes

    // block -> (pair, image, head, query block, key split): attention.hip's XCD-aware map
    const int gsz = (a.batch % 2 == 0) ? 2 : a.batch;
    const int hz = a.heads * gsz;
    const int nqb = (a.n_max + 127) / 128;
    const int per_group = hz * nqb * (a.part ? ATTN_MAX_SPLIT : 1);
    const int grp_idx = blockIdx.x / per_group;
    const int bid = blockIdx.x - grp_idx * per_group;
    const int head = (bid % hz) % a.heads, z = grp_idx * gsz + (bid % hz) / a.heads;
    const int y = a.cross ? (z ^ 1) : z;
    if (a.active && a.active[(z >> 1) * a.pstride] == 0) return;
    const int nq = a.n_ptr ? a.n_ptr[(z >> 1) * a.pstride + (z & 1)] : a.n_max;
    const int nk_all = a.n_ptr ? a.n_ptr[(y >> 1) * a.pstride + (y & 1)] : a.n_max;
    const int qblk = (bid / hz) % nqb, split = (bid / hz) / nqb;
    const int qb = qblk * 128;
    if (qb >= nq || nk_all <= 0) return;
    int n_split = 1, k0 = 0, nk = nk_all;
    if (a.part) {       // split-KV, decided on the device (attention.hip): few live queries -> the keys of a query block on up to 4 blocks
        const int steps_all = (nk_all + BKT - 1) / BKT;
        const int want = nq > 2048 ? 1 : (nq > 1024 ? 2 : ATTN_MAX_SPLIT);
        const int steps_per = (steps_all + want - 1) / want;
        n_split = (steps_all + steps_per - 1) / steps_per;
        k0 = split * steps_per * BKT;
        nk = min(nk_all - k0, steps_per * BKT);
    }
    if (split >= n_split) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const int qrow = qb + wave * 32 + c;

    const float* Q = a.q + (long)z * a.bstride + (long)head * a.hstride;
    __amdgpu_buffer_rsrc_t K, V;
    du32x4 Kd = {}, Vd = {};
    if constexpr (PRE) {
        const long rows = (long)a.heads * a.n_max * a.batch;
        const unsigned char* kp = reinterpret_cast<const unsigned char*>(a.planes) + (((long)y * a.heads + head) * a.n_max + k0) * 384;
        K = bx_rsrc_bytes(kp, nk * 384);
        V = bx_rsrc_bytes(kp + rows * 384, nk * 384);
        if constexpr (DMA) { Kd = dma_rsrc(kp, nk * 384); Vd = dma_rsrc(kp + rows * 384, nk * 384); }
    } else {
        K = bx_rsrc_bytes(a.k + (long)y * a.bstride + (long)head * a.hstride + (long)k0 * 64, nk * 256);
        V = bx_rsrc_bytes(a.v + (long)y * a.bstride + (long)head * a.hstride + (long)k0 * 64, nk * 256);
    }

    // Q planes: lane (c, hh) keeps Q[qrow][16 s + 8 hh + j], j = 0..7, of d-chunk s; the softmax scale and log2 e folded in before the cut
    u32x4 qh[4], qm[4], ql[4];
    {
        const float* qp = Q + (long)min(qrow, nq - 1) * 64 + hh * 8;
        const float c2 = a.scale * 1.4426950408889634f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float4 u = *reinterpret_cast<const float4*>(qp + 16 * s), w = *reinterpret_cast<const float4*>(qp + 16 * s + 4);
            const Planes p = split8(u.x * c2, u.y * c2, u.z * c2, u.w * c2, w.x * c2, w.y * c2, w.z * c2, w.w * c2);
            qh[s] = p.h; qm[s] = p.m; ql[s] = p.l;
        }
    }

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    const int nt = (nk + BKT - 1) / BKT;

    // LDS: K ring (2 x 3 planes), V ring (2 x 3 planes); K runs one step ahead of V
    unsigned char* const kring = bsm;
    unsigned char* const vring = bsm + RD * 3 * KPL;
    // fragment reads. K: row c, d-chunk s, pieces 2 s + hh of the row. V (transposed read): row 16 s + 8 half + 4 hh + q, bytes 64 dt + 32 dgrp + 8 pp
    int kq[4], dv[2];
    const int vq = (lane & 15) >> 2;
    if constexpr (DMA) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) kq[s4] = c * 128 + (((2 * s4 + hh) ^ ((c >> 1) & 7)) << 4);
        const int base = (4 * hh + vq) * 128 + (((lane >> 4) & 1) * 2 + ((lane & 3) >> 1)) * 16 + (lane & 1) * 8, sv = (vq >> 1) & 1;
        dv[0] = base + sv * 64; dv[1] = base + (1 - sv) * 64;
    } else {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) kq[s4] = c * BKS + hh * 16 + 32 * s4;
        const int base = (4 * hh + vq) * BVS + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
        dv[0] = base; dv[1] = base + 64;
    }

    // staging. PRE: a tile is 32 rows x 384 bytes = 768 16-byte pieces, contiguous in memory: piece id = tid + 256 i sits at byte 16 id,
    // row id / 24, plane (id % 24) >> 3, 16-byte column id & 7. !PRE: rows tid >> 4 and + 16, floats 4 (tid & 15) .. + 3.
    u32x4 pk[3], pv[3];
    float4 fk[2], fv[2];
    int lk[3], lv[3];
    if constexpr (PRE) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int id = tid + 256 * i, row = id / 24, rem = id - 24 * row;
            lk[i] = (rem >> 3) * BK_PLANE + row * BKS + (rem & 7) * 16;
            lv[i] = (rem >> 3) * BV_PLANE + row * BVS + (rem & 7) * 16;
        }
    }
    const unsigned voff = PRE ? tid * 16u : ((tid >> 4) * 64 + (tid & 15) * 4) * 4u;
    const int st_k = (tid >> 4) * BKS + (tid & 15) * 8, st_v = (tid >> 4) * BVS + (tid & 15) * 8;
    auto load_k = [&](int tile) {
        if constexpr (PRE) {
            const unsigned so = (unsigned)tile * (BKT * 384u);
#pragma unroll
            for (int i = 0; i < 3; ++i) pk[i] = __builtin_amdgcn_raw_buffer_load_b128(K, voff, so + 4096u * i, 0);
        } else {
            const unsigned so = (unsigned)tile * (BKT * 256u);
            fk[0] = bx_load4(K, voff, so); fk[1] = bx_load4(K, voff, so + 16 * 256u);
        }
    };
    auto load_v = [&](int tile) {
        if constexpr (PRE) {
            const unsigned so = (unsigned)tile * (BKT * 384u);
#pragma unroll
            for (int i = 0; i < 3; ++i) pv[i] = __builtin_amdgcn_raw_buffer_load_b128(V, voff, so + 4096u * i, 0);
        } else {
            const unsigned so = (unsigned)tile * (BKT * 256u);
            fv[0] = bx_load4(V, voff, so); fv[1] = bx_load4(V, voff, so + 16 * 256u);
        }
    };
    auto store_k = [&](int stage) {
        unsigned char* const kp = kring + stage * 3 * BK_PLANE;
        if constexpr (PRE) {
#pragma unroll
            for (int i = 0; i < 3; ++i) *reinterpret_cast<u32x4*>(kp + lk[i]) = pk[i];
        } else {
            stage4(kp, BK_PLANE, st_k, fk[0]); stage4(kp, BK_PLANE, st_k + 16 * BKS, fk[1]);
        }
    };
    auto store_v = [&](int stage) {
        unsigned char* const vp = vring + stage * 3 * BV_PLANE;
        if constexpr (PRE) {
#pragma unroll
            for (int i = 0; i < 3; ++i) *reinterpret_cast<u32x4*>(vp + lv[i]) = pv[i];
        } else {
            stage4(vp, BV_PLANE, st_v, fv[0]); stage4(vp, BV_PLANE, st_v + 16 * BVS, fv[1]);
        }
    };
    // DMA form: wave w transfers rows 8 w .. 8 w + 7 of the three planes of a tile (three 1 KB pieces per tensor), lane (row, place pc) fetching piece
    // pc ^ swz(row) of its row
    const int drow = 8 * wave + (lane >> 3), dpc = lane & 7;
    const unsigned dvoff_k = drow * 384u + ((dpc ^ ((drow >> 1) & 7)) << 4), dvoff_v = drow * 384u + ((dpc ^ (((drow >> 1) & 1) << 2)) << 4);
    const unsigned wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);          // in an SGPR: the transfer's LDS address goes through M0
    const unsigned lds_k = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(dlds_ptr_t)kring) + wave_u * 1024u;
    const unsigned lds_v = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(dlds_ptr_t)vring) + wave_u * 1024u;
    auto dma_k = [&](int tile, int stage) {
#pragma unroll
        for (int i = 0; i < 3; ++i) bx_dma16(Kd, __builtin_amdgcn_readfirstlane(lds_k + (unsigned)stage * (3u * DPL) + i * DPL), dvoff_k, __builtin_amdgcn_readfirstlane((unsigned)tile * (BKT * 384u) + i * 128u));
    };
    auto dma_v = [&](int tile, int stage) {
#pragma unroll
        for (int i = 0; i < 3; ++i) bx_dma16(Vd, __builtin_amdgcn_readfirstlane(lds_v + (unsigned)stage * (3u * DPL) + i * DPL), dvoff_v, __builtin_amdgcn_readfirstlane((unsigned)tile * (BKT * 384u) + i * 128u));
    };
    // S^T = K . Q^T of one tile: two accumulators (even / odd d-chunks), the small products first
    auto qk = [&](int stage, f32x16& out) {
        const unsigned char* const kp = kring + stage * 3 * KPL;
        f32x16 xa, xb;
#pragma unroll
        for (int r = 0; r < 16; ++r) { xa[r] = 0.f; xb[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const u32x4 kh = *reinterpret_cast<const u32x4*>(kp + kq[s]);
            const u32x4 km = *reinterpret_cast<const u32x4*>(kp + KPL + kq[s]);
            const u32x4 kl = *reinterpret_cast<const u32x4*>(kp + 2 * KPL + kq[s]);
            f32x16& x = (s & 1) ? xb : xa;
            x = mfma_bf(kh, ql[s], x);
            x = mfma_bf(kl, qh[s], x);
            x = mfma_bf(km, qm[s], x);
            x = mfma_bf(kh, qm[s], x);
            x = mfma_bf(km, qh[s], x);
            x = mfma_bf(kh, qh[s], x);
        }
        out = xa + xb;
    };
    // O^T += V^T . P^T: registers 8 s .. 8 s + 7 of P are the eight k slots of key chunk s
    auto pv_mul = [&](int stage, const f32x16& p) {
        const unsigned char* const vp = vring + stage * 3 * VPL;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const Planes pp = split8(p[8 * s + 0], p[8 * s + 1], p[8 * s + 2], p[8 * s + 3], p[8 * s + 4], p[8 * s + 5], p[8 * s + 6], p[8 * s + 7]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const unsigned char* vb = vp + (16 * s) * VS + dv[dt];
                const u32x2 h0 = tr_read(vb), h1 = tr_read(vb + 8 * VS);
                const u32x2 m0 = tr_read(vb + VPL), m1 = tr_read(vb + VPL + 8 * VS);
                const u32x2 l0 = tr_read(vb + 2 * VPL), l1 = tr_read(vb + 2 * VPL + 8 * VS);
                const u32x4 vh = {h0.x, h0.y, h1.x, h1.y}, vm = {m0.x, m0.y, m1.x, m1.y}, vl = {l0.x, l0.y, l1.x, l1.y};
                f32x16& o = dt ? o1 : o0;
                o = mfma_bf(vh, pp.l, o);
                o = mfma_bf(vl, pp.h, o);
                o = mfma_bf(vm, pp.m, o);
                o = mfma_bf(vh, pp.m, o);
                o = mfma_bf(vm, pp.h, o);
                o = mfma_bf(vh, pp.h, o);
            }
        }
    };

    // ---- prologue: K0 -> LDS, S(0); K1, V0 -> LDS; K2, V1 in registers
    f32x16 sc, sn;
    if constexpr (DMA) {       // tile j lives in ring stage j % 3
        dma_k(0, 0); dma_k(1, 1); dma_v(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        dma_k(2, 2); dma_v(1, 1);              // in flight under S(0); waited for at the end of step 0
        qk(0, sc);
        __syncthreads();                       // every wave is done with K(0) before step 0 transfers K(3) over it
    } else {
        load_k(0);
        store_k(0);
        load_k(1);
        load_v(0);
        __syncthreads();
        qk(0, sc);
        store_k(1);
        store_v(0);
        load_k(2);
        load_v(1);
        __syncthreads();
    }

    // step t: stage K(t+2), V(t+1); request K(t+3), V(t+2); reference of tile t; then 48 MFMA slots - S(t+1) = 24, O += P(t) V(t) = 24 -
    // with the vector work of tile t dealt over them BY HAND, a few instructions behind each MFMA and a scheduling fence behind each slot:
    // the bf16 MFMA holds the issue port for 8 of its 32 cycles, so ~5 vector instructions per slot are free, but only if they sit
    // between the MFMAs in program order (left to itself the compiler emits the MFMAs in runs of six and the vector work in runs of 60).
    //   slots  0..23 (S(t+1), d-chunk s = slot / 6):  exp2 + row sum of P registers 0..7 | cut of key chunk 0 | exp2 + row sum of 8..15
    //   slots 24..35 (O += V P, key chunk 0):          cut of key chunk 1
    //   slots 36..47 (key chunk 1):                    nothing left
    // LDS reads run one group ahead: K fragments of d-chunk s + 1 during chunk s, the six transposed V reads of (chunk, d-tile) group g + 1
    // during group g.
    static constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};   // the six products, small ones first: A plane, B plane
    float m_use = 0.f;
    if (nt >= 2) m_use = softmax_ref<false>(sc, 0, nk, hh, m_run, l_run, o0, o1);     // tile 0 lies inside the keys
    int r0 = 0, r1 = 1 % RD, r2 = 2 % RD;    // t % RD, (t + 1) % RD, (t + 2) % RD
    // im_debug_clock_probe (a.clock set only by that call): the shader clock this kernel holds inside its main loop = cycles / 100 MHz ticks
    unsigned long long ck0 = 0, cr0 = 0;
    if (a.clock) { ck0 = __builtin_amdgcn_s_memtime(); cr0 = __builtin_amdgcn_s_memrealtime(); }
    for (int t = 0; t < nt - 2; ++t) {       // tiles t and t + 1 lie inside the keys: no masks anywhere
        const unsigned char* const kp = kring + r1 * 3 * KPL;
        const unsigned char* const vp = vring + r0 * 3 * VPL;
        u32x4 kf[2][3];
        u32x2 vr[2][6];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) kf[0][pl] = kf_read(kp + pl * KPL + kq[0]);
#ifndef BX_ABL_NO_STAGE     // -DBX_ABL_*: timing-only ablations of the main loop (wrong results), as IM_ABL_* in attention.hip
        if constexpr (DMA) {
            dma_k(t + 3, r0);                // over K(t), last read in step t - 1
            dma_v(t + 2, r2);                // over V(t - 1)
        } else {
            store_k(r0);
            store_v(r1);
            load_k(t + 3);
            load_v(t + 2);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);

        float ra[8], rb[8], rsum0 = 0.f, rsum1 = 0.f;
        unsigned ch[8], cm[8], cl[8];
        // An empty asm with the values as read-write operands in front of and behind every piece: the pieces cannot leave their slot (the
        // scheduling fences alone hold the MFMAs in place, but instruction selection had moved the cuts behind the last MFMA of the phase).
        // No instruction is emitted, so nothing is hidden from the hazard recogniser.
#define BX_PIN(x) asm volatile("" : "+v"(x))
        auto expo = [&](int r) {                       // 3 instructions
#ifdef BX_ABL_NO_EXPCUT     // timing ablation (wrong results): no exp2, no row sums, no plane cut of P
            return;
#endif
            float x = sc[r];
            BX_PIN(x);
            x = __builtin_amdgcn_exp2f(x - m_use);
            BX_PIN(x);
            sc[r] = x;
            if (r & 1) rsum1 += x; else rsum0 += x;
        };
        auto cut_a = [&](int i) {                      // pair i = registers 2 i, 2 i + 1: 4-5 instructions
#ifdef BX_ABL_NO_EXPCUT
            ch[i] = __float_as_uint(sc[2 * i]); cm[i] = __float_as_uint(sc[2 * i + 1]); cl[i] = ch[i]; return;
#endif
            float x = sc[2 * i], y = sc[2 * i + 1];
            BX_PIN(x); BX_PIN(y);
            ch[i] = cvt_pk(x, y);
            ra[i] = x - __uint_as_float(ch[i] << 16);
            rb[i] = y - __uint_as_float(ch[i] & 0xffff0000u);
            BX_PIN(ra[i]); BX_PIN(rb[i]);
        };
        auto cut_b = [&](int i) {
#ifdef BX_ABL_NO_EXPCUT
            return;
#endif
            BX_PIN(ra[i]); BX_PIN(rb[i]);
            cm[i] = cvt_pk(ra[i], rb[i]);
            ra[i] -= __uint_as_float(cm[i] << 16);
            rb[i] -= __uint_as_float(cm[i] & 0xffff0000u);
            BX_PIN(ra[i]); BX_PIN(rb[i]);
        };
        auto cut_l = [&](int i) {
#ifdef BX_ABL_NO_EXPCUT
            return;
#endif
            BX_PIN(ra[i]);
            cl[i] = cvt_pk(ra[i], rb[i]);
            BX_PIN(cl[i]);
        };

        // ---- slots 0..23: S(t+1) = K(t+1) . Q^T
        f32x16 xa, xb;
#pragma unroll
        for (int r = 0; r < 16; ++r) { xa[r] = 0.f; xb[r] = 0.f; }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            if (s4 < 3) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) kf[(s4 + 1) & 1][pl] = kf_read(kp + pl * KPL + kq[s4 + 1]);
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int slot = 6 * s4 + j;
                const u32x4 qb = PB[j] == 0 ? qh[s4] : (PB[j] == 1 ? qm[s4] : ql[s4]);
                if (j & 1) xb = mfma_bf(kf[s4 & 1][PA[j]], qb, xb);
                else xa = mfma_bf(kf[s4 & 1][PA[j]], qb, xa);
                if (slot < 8) expo(slot);
                else if (slot < 16) { if (slot & 1) cut_b((slot - 8) >> 1); else cut_a((slot - 8) >> 1); }
                else { expo(slot - 8); if (slot < 20) cut_l(slot - 16); }
                if (slot >= 18) vr[0][slot - 18] = tr_read(vp + dv[0] + ((slot - 18) >> 1) * VPL + ((slot - 18) & 1) * 8 * VS);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- slots 24..47: O^T += V(t)^T . P(t)^T, groups g = (key chunk g >> 1, d-tile g & 1)
        float mxn = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ks = g >> 1;
            const u32x4 pp[3] = {u32x4{ch[4 * ks], ch[4 * ks + 1], ch[4 * ks + 2], ch[4 * ks + 3]},
                                 u32x4{cm[4 * ks], cm[4 * ks + 1], cm[4 * ks + 2], cm[4 * ks + 3]},
                                 u32x4{cl[4 * ks], cl[4 * ks + 1], cl[4 * ks + 2], cl[4 * ks + 3]}};
            const u32x4 vf[3] = {u32x4{vr[g & 1][0].x, vr[g & 1][0].y, vr[g & 1][1].x, vr[g & 1][1].y},
                                 u32x4{vr[g & 1][2].x, vr[g & 1][2].y, vr[g & 1][3].x, vr[g & 1][3].y},
                                 u32x4{vr[g & 1][4].x, vr[g & 1][4].y, vr[g & 1][5].x, vr[g & 1][5].y}};
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int slot = 6 * g + j;
                if (g & 1) o1 = mfma_bf(vf[PA[j]], pp[PB[j]], o1);
                else o0 = mfma_bf(vf[PA[j]], pp[PB[j]], o0);
                if (slot < 8) { if (slot & 1) cut_b(4 + (slot >> 1)); else cut_a(4 + (slot >> 1)); }
                else if (slot < 12) cut_l(4 + slot - 8);
                if (slot >= 8 && slot < 16) {                     // S(t+1) = the two accumulators, two registers per slot
                    float u0 = xa[2 * (slot - 8)] + xb[2 * (slot - 8)], u1 = xa[2 * (slot - 8) + 1] + xb[2 * (slot - 8) + 1];
                    BX_PIN(u0); BX_PIN(u1);
                    sn[2 * (slot - 8)] = u0; sn[2 * (slot - 8) + 1] = u1;
                }
                if (slot >= 16 && slot < 24) {                    // its row maximum, one v_max3 per slot
                    const int i = slot - 16;
                    float u = i == 0 ? bmax3(sn[0], sn[1], sn[2]) : (i == 7 ? fmaxf(mxn, sn[15]) : bmax3(mxn, sn[2 * i + 1], sn[2 * i + 2]));
                    BX_PIN(u);
                    mxn = u;
                }
                if (g < 3) {
                    const int gn = g + 1;
                    vr[gn & 1][j] = tr_read(vp + (16 * (gn >> 1)) * VS + dv[gn & 1] + (j >> 1) * VPL + (j & 1) * 8 * VS);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        float rs = rsum0 + rsum1;
        rs += __shfl_xor(rs, 32);
        l_run += rs;
        // the reference of tile t + 1, behind the last MFMA that accumulates into O with the old one
        mxn = fmaxf(mxn, __shfl_xor(mxn, 32));
        if (__builtin_amdgcn_ballot_w64(mxn > m_run + BX_SLACK) != 0) {
            asm volatile("" ::: "memory");
            const float m_new = fmaxf(m_run, mxn);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
            o0 *= alpha;
            o1 *= alpha;
            m_run = m_new;
        }
        m_use = m_run;
        sc = sn;
        { const int rr = r0; r0 = r1; r1 = r2; r2 = RD == 3 ? rr : r0; }
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // K(t+2), V(t+1) have landed; this step's six transfers may still fly
#ifndef BX_ABL_NO_BARRIER
        __syncthreads();
#endif
    }
#undef BX_PIN
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // nothing is requested from here on
    if (nt >= 2) {                           // step nt - 2 in plain form: its look-ahead tile is the last one, whose reference needs the mask
        if constexpr (DMA) {
            __syncthreads();                 // the other waves' transfers too
        } else {
            store_k(r0);
            store_v(r1);
        }
        qk(r1, sn);
        softmax_exp(sc, m_use, l_run);
        pv_mul(r0, sc);
        sc = sn;
        { const int rr = r0; r0 = r1; r1 = r2; r2 = RD == 3 ? rr : r0; }
        __syncthreads();
    }
    if (a.clock && nt > 8) {                 // blocks with a real loop only; the words go nowhere else
        const unsigned long long ck1 = __builtin_amdgcn_s_memtime(), cr1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) {
            unsigned long long* cp = a.clock + 2 * (blockIdx.x % CLOCK_PROBE_SLOTS);
            cp[0] = ck1 - ck0; cp[1] = cr1 - cr0;
        }
    }
    {
        const int t = nt - 1;
        m_use = softmax_ref<true>(sc, t * BKT, nk, hh, m_run, l_run, o0, o1);
        softmax_exp(sc, m_use, l_run);
        pv_mul(r0, sc);
    }

    // ---- split-KV: park the partial (O, m, l); the last block of this query block merges all of them in split order (attention.hip)
    if (n_split > 1) {
        const long prow = (((long)z * a.heads + head) * a.n_max + min(qrow, a.n_max - 1)) * 66;
        const long pstride = (long)a.batch * a.heads * a.n_max * 66;
        float* pp = a.part + (long)split * pstride + prow;
        if (qrow < nq) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *reinterpret_cast<float2*>(pp + 8 * g + 4 * hh) = make_float2(o0[4 * g], o0[4 * g + 1]);
                *reinterpret_cast<float2*>(pp + 8 * g + 4 * hh + 2) = make_float2(o0[4 * g + 2], o0[4 * g + 3]);
                *reinterpret_cast<float2*>(pp + 32 + 8 * g + 4 * hh) = make_float2(o1[4 * g], o1[4 * g + 1]);
                *reinterpret_cast<float2*>(pp + 32 + 8 * g + 4 * hh + 2) = make_float2(o1[4 * g + 2], o1[4 * g + 3]);
            }
            if (hh == 0) { pp[64] = m_run; pp[65] = l_run; }
        }
        __threadfence();
        __syncthreads();
        int* flag = reinterpret_cast<int*>(bsm);
        if (tid == 0) {
            int* cnt = a.counters + ((long)z * a.heads + head) * nqb + qblk;
            const int old = atomicAdd(cnt, 1);
            *flag = (old == n_split - 1);
            if (old == n_split - 1) *cnt = 0;            // ready for the next launch
        }
        __syncthreads();
        if (!*flag) return;
        __threadfence();
        float m = -INFINITY;
        for (int sidx = 0; sidx < n_split; ++sidx) m = fmaxf(m, a.part[(long)sidx * pstride + prow + 64]);
        l_run = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
        for (int sidx = 0; sidx < n_split; ++sidx) {
            const float* ps = a.part + (long)sidx * pstride + prow;
            const float w = __builtin_amdgcn_exp2f(ps[64] - m);
            l_run += w * ps[65];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    o0[4 * g + q] += w * ps[8 * g + 4 * hh + q];
                    o1[4 * g + q] += w * ps[32 + 8 * g + 4 * hh + q];
                }
        }
    }

    // ---- epilogue: lane (c, hh) holds query qrow, d = 32 dt + 8 g + 4 hh + (0..3) in registers 4g..4g+3 of o<dt>
    if (qrow < nq) {
        const float inv = 1.f / l_run;
        float* op = a.out + (long)z * a.out_bstride + (long)qrow * a.ldo + head * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 w0 = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            const float4 w1 = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
            *reinterpret_cast<float4*>(op + 8 * g + 4 * hh) = w0;
            *reinterpret_cast<float4*>(op + 32 + 8 * g + 4 * hh) = w1;
        }
    }
}

// K / V of the launch as bf16 triples (a.planes); a launch of its own so that event profiles and rocprofv3 name the two kernels apart
hipError_t launch_attn_planes(const AttnArgs& a, hipStream_t s) {
    static const bool f32_form = [] { const char* e = getenv("IM_ATTN_F32"); return e && atoi(e) != 0; }();
    if (!a.planes || a.n_max <= 0 || f32_form || a.f32_form) return hipSuccess;
    const long rows = (long)a.batch * a.heads * a.n_max;
    hipLaunchKernelGGL(kv_planes_kernel, dim3((unsigned)((rows + 31) / 32), 2), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_flash_attn_bx(const AttnArgs& a, hipStream_t s) {
    static size_t lds_pre[IM_MAX_DEVICES] = {0}, lds_cut[IM_MAX_DEVICES] = {0};
    dim3 grid(((a.n_max + 127) / 128) * a.heads * a.batch * (a.part ? ATTN_MAX_SPLIT : 1)), block(256);
    static size_t lds_dma[IM_MAX_DEVICES] = {0};
    static const bool reg_staging = [] { const char* e = getenv("IM_ATTN_REG_STAGING"); return e && atoi(e) != 0; }();
    if (a.planes && !reg_staging) {     // planes filled by launch_attn_planes on the same stream; tiles into LDS by LDS-DMA
        if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(&flash_attn_bx_kernel<true, true>), D_LDS, lds_dma); e != hipSuccess) return e;
        hipLaunchKernelGGL((flash_attn_bx_kernel<true, true>), grid, block, D_LDS, s, a);
    } else if (a.planes) {              // tiles through registers (A/B)
        if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(&flash_attn_bx_kernel<true, false>), B_LDS, lds_pre); e != hipSuccess) return e;
        hipLaunchKernelGGL((flash_attn_bx_kernel<true, false>), grid, block, B_LDS, s, a);
    } else {
        if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(&flash_attn_bx_kernel<false, false>), B_LDS, lds_cut); e != hipSuccess) return e;
        hipLaunchKernelGGL((flash_attn_bx_kernel<false, false>), grid, block, B_LDS, s, a);
    }
    return hipGetLastError();
}

}  // namespace im
