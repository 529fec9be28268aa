// Shared pieces of the Winograd F(2x2, 3x3) kernels (conv_wino.hip: four waves per block, two blocks per CU; conv_wino_bx2.hip: eight waves,
// two tile groups that share every U fragment through LDS): vector helpers, LDS-DMA, the bf16 plane cut and the inverse-transform epilogue.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "sp_post.h"

namespace im {

static constexpr int WCC = 8;                         // channels per slab
static constexpr int W_SU = 16 * 64 * WCC;            // floats: one U block

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wmake_rsrc(const void* base, unsigned bytes) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    void* p = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
// packed fp32 on float4 halves: two-element vector arithmetic, which the compiler lowers to v_pk_add_f32 (plain float4 arithmetic
// is scalarised). NOT inline asm: round 3 found that `asm("v_pk_add_f32 ...")` next to the MFMAs gives wrong results as soon as
// the register allocation changes (any reordering of the slab step that keeps the U fragments live across the stage; the same
// source with these compiler-visible adds is correct, tools/conv_reorder_test.py) - the hazard recogniser cannot see into an asm
// statement, so a VALU write that an MFMA reads too early goes unprotected.
__device__ __forceinline__ float4 add4(float4 x, float4 y) {
    const f32x2 a = {x.x, x.y}, b = {x.z, x.w}, c = {y.x, y.y}, d = {y.z, y.w};
    const f32x2 r0 = a + c, r1 = b + d;
    return make_float4(r0.x, r0.y, r1.x, r1.y);
}
// x - y as fma(y, -1, x) with the -1 in a register the compiler cannot see through (`minus_one()`): the same value, rounded once,
// but lowered to v_pk_fma_f32, where a vector subtraction (and an fma by a literal -1, which is folded back into one) is scalarised
// into two v_sub_f32. fp32 MFMA and VALU never co-execute on this part, so every vector instruction saved is MFMA time.
// The asm is NOT volatile: a volatile asm statement counts as a possible store, after which the compiler no longer proves the
// uniform weight loads of the fused first layer unclobbered and turns its scalar loads into 20 vector loads per stage (+15 %).
__device__ __forceinline__ f32x2 minus_one() {
    float m = -1.f;
    asm("" : "+s"(m));
    return f32x2{m, m};
}
__device__ __forceinline__ float4 sub4(float4 x, float4 y, f32x2 m1) {
    const f32x2 a = {x.x, x.y}, b = {x.z, x.w}, c = {y.x, y.y}, d = {y.z, y.w};
    const f32x2 r0 = __builtin_elementwise_fma(c, m1, a), r1 = __builtin_elementwise_fma(d, m1, b);
    return make_float4(r0.x, r0.y, r1.x, r1.y);
}
// max(v, floor) as one v_med3_f32 (fmaxf of two values of unknown origin costs two canonicalising v_max besides the max itself)
__device__ __forceinline__ f32x16 sub16(const f32x16& x, const f32x16& y, f32x2 m1) {
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        const f32x2 v = __builtin_elementwise_fma(f32x2{y[i], y[i + 1]}, m1, f32x2{x[i], x[i + 1]});
        r[i] = v.x; r[i + 1] = v.y;
    }
    return r;
}

static constexpr int S_TH = 8, S_TW = 16;
static constexpr int S_PH = S_TH + 2, S_PW = S_TW + 2;
static constexpr int S_ROW = 20, S_PAR = 10, S_QUAD = S_PH * S_ROW;   // slots
static constexpr int S_SP = 2 * S_QUAD * 4;                 // floats per patch stage
static constexpr int S_MAIN = 2 * S_SP + 2 * W_SU;
static constexpr int S_IH = S_TH + 4, S_IW = S_TW + 4;
static constexpr int S_FUSE = S_IH * S_IW;
static constexpr int S_LDS_FLOATS = S_MAIN;
// BX, plain layers: the patch stage of a 16-channel chunk is PIXEL-major - one (patch row, column parity) = 10 slots x 4 channel quads = 640 bytes,
// filled by ONE LDS-DMA piece whose lanes 4 s .. 4 s + 3 read the four quads of pixel slot s (64 contiguous bytes of global memory: a piece touches
// ~10 cache lines where the quad-major pieces of the f32 form touch 64, and the texture-address unit is what these kernels are bound by,
// profiles/r06_conv_bx_ablations.txt). Reads stay conflict-free: the 16 lanes a ds_read_b128 serves together are 4 tile columns (64 bytes apart)
// x 4 tile rows, and the rows of a row pair start 16 bytes further per pair (XP_PITCH leaves room for the shift).
static constexpr int XP_PITCH = 44;                  // float4 per (row, parity): 10 slots x 4 quads + up to 3 of shift, padded to a multiple of 4
static constexpr int XP_STAGE = 2 * S_PH * XP_PITCH; // float4 per stage: 880 = 14,080 bytes
static constexpr int X_SP = XP_STAGE * 4;            // BX: floats per patch stage (16 channels)
static constexpr int X_LDS_FLOATS = 16384;           // BX: two patch stages (6,400 floats); the 64 KB exchange image of the epilogue aliases them
static_assert(2 * X_SP <= X_LDS_FLOATS, "the patch stages must fit under the exchange image");

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// One LDS-DMA piece (buffer_load_dwordx4 ... lds: 64 lanes x 16 bytes land at lds_byte_addr + 16 * lane) issued as INLINE ASM, so
// that the compiler's wait-count pass does not know about it: with the `__builtin_amdgcn_raw_ptr_buffer_load_lds` form it orders
// every later LDS read behind the transfer with `s_waitcnt vmcnt(0)` (it cannot tell which LDS bytes a transfer writes), i.e. each
// slab step first waited for the transfer of the NEXT slab that it had just started. Issued this way the transfer of slab + 1 stays
// in flight under the reads, the transform and the 32 MFMAs of slab and is waited for by an explicit s_waitcnt vmcnt(0) in front
// of the step's barrier (IM_DMA_WAIT). The descriptor is four SGPRs, the LDS address goes through M0.
typedef unsigned int wu32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ wu32x4 wmake_rsrc4(const void* base, unsigned bytes) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    wu32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) & 0xFFFFu;
    r.z = __builtin_amdgcn_readfirstlane(bytes);
    r.w = 0x00020000u;
    return r;
}
// M0 is named in the clobber list so that a compiler-generated M0 user (builtin LDS-DMA, readlane / movrel, sendmsg) placed in the
// same kernel never relies on a value from before the statement; clang notes that M0 is a reserved register (-Winline-asm), which is
// the point: silenced for this function only.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16(wu32x4 rsrc, unsigned lds_byte_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_byte_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
#pragma clang diagnostic pop
#define IM_DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// the same when eight register loads (the next slab's U fragments) were issued BEHIND the transfers: loads return in order, so "at most
// eight outstanding" means the transfers have landed while the U loads stay in flight across the barrier (round 5)
#define IM_DMA_WAIT_BEFORE_8_LOADS() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")

// ---- BX (round 6): the sixteen element-wise products on the bf16 matrix cores at fp32 accuracy, as gemm.hip BX / attention_bx.hip: every fp32
// operand is the exact sum of three bf16 values (x = h + m + l, each rounded to nearest even from the residual), a product is the six bf16
// products h l, l h, m m, h m, m h, h h accumulated in fp32 in that order (small terms first). U is cut on the host (weights.hip::
// pack_conv3x3_wino_bx); V = B^T d B is computed in fp32 exactly as in the f32 form and cut AFTER the transform (the planes of a sum are not
// the sums of the planes), in registers, just before it becomes the A operand of v_mfma_f32_32x32x16_bf16 (lane (c, hh): tile c, channels
// 8 hh .. 8 hh + 7 of a 16-channel chunk).
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wbf16x2 __attribute__((ext_vector_type(2)));
// -DIM_XABL_*: timing-only ablations of the BX main loop (WRONG results by construction; tools/build_conv_variant.sh builds them into build_abl/)
__device__ __forceinline__ f32x16 mfma_bx(wu32x4 a, wu32x4 b, f32x16 c) {
#ifdef IM_XABL_NO_MFMA
    c[0] += __uint_as_float(a.x ^ b.x);      // keeps the operands alive: one vector instruction instead of the MFMA
    return c;
#endif
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wbf16x8, a), __builtin_bit_cast(wbf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned wcvt_pk(float a, float b) {
    const wbf16x2 v = __builtin_convertvector(f32x2{a, b}, wbf16x2);
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void wsplit2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
#ifdef IM_XABL_NO_CUT
    h = __float_as_uint(a); m = __float_as_uint(b); l = h ^ m;
    return;
#endif
    h = wcvt_pk(a, b);
    float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = wcvt_pk(ra, rb);
    ra -= __uint_as_float(m << 16);
    rb -= __uint_as_float(m & 0xffff0000u);
    l = wcvt_pk(ra, rb);
}
struct WPlanes { wu32x4 h, m, l; };
__device__ __forceinline__ WPlanes wsplit8(const float (&x)[8]) {
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wsplit2(x[2 * i], x[2 * i + 1], h[i], m[i], l[i]);
    return WPlanes{wu32x4{h[0], h[1], h[2], h[3]}, wu32x4{m[0], m[1], m[2], m[3]}, wu32x4{l[0], l[1], l[2], l[3]}};
}
#ifdef IM_XABL_ONE_U          // every step reads the same 3 KB fragment of its wave (L1-resident): what is the L2 stream of the U planes worth?
#define IM_XABL_U_OFFSET(x) (ux_base + 0u * (x))
#elif defined(IM_XABL_NO_U)   // beyond the descriptor's range: zeros, no traffic
#define IM_XABL_U_OFFSET(x) (0u * (x))
#else
#define IM_XABL_U_OFFSET(x) (x)
#endif
static constexpr int X_RING = 4;                     // register slots of U fragments (one (position, 32 output channels) fragment = 3 planes x 4 registers)
static constexpr int X_AHEAD = X_RING - 1;           // fragments requested ahead of the one in use


// The end of a block of four waves (wave ph = V row ph of 32 tiles x 64 output channels; conv_wino.hip's kernels and the groups of conv_wino_bx2.hip):
// xw = the group's 64 KB exchange image in LDS (nothing else of the group may live there any more), b / y0 / x0 / co0 = image, region origin and
// first output channel. Contains ONE __syncthreads(): every wave of the block must call it together.
template <bool POOL>
__device__ __forceinline__ void wino_epilogue(const ConvArgs& a, const f32x16 (&acc)[8], float4* const xw, const int ph, const int lane, const int b,
                                              const int y0, const int x0, const int co0, const f32x2 m1) {
    const int c = lane & 31, hh = lane >> 5;
    // ---- inverse transform Y = A^T M A. acc[2 j + n]: position (ph, j), output channels n * 32 + c. Pass over j in registers:
    //   s_ph[0] = (M0 + M1) + M2,  s_ph[1] = (M1 - M2) - M3;   then over the four V rows = the four waves:
    //   Y[0][b] = (s0[b] + s1[b]) + s2[b],   Y[1][b] = s1[b] - (s2[b] + s3[b])       (the associations of the two-row form: same bits)
    // Wave w finishes tile row w (accumulator registers 4 w .. 4 w + 3 of every s): each wave hands the other three the four registers
    // they own of its four s vectors ([source][owner][b][n][lane] float4: 12 writes, 12 reads per lane, one barrier).
    f32x16 sv[2][2];     // [b][n]
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        sv[0][n] = (acc[0 + n] + acc[2 + n]) + acc[4 + n];
        sv[1][n] = sub16(sub16(acc[2 + n], acc[4 + n], m1), acc[6 + n], m1);
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w != ph) {                                       // wave-uniform
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    xw[(((ph * 4 + w) * 2 + bb) * 2 + n) * 64 + lane] =
                        make_float4(sv[bb][n][4 * w], sv[bb][n][4 * w + 1], sv[bb][n][4 * w + 2], sv[bb][n][4 * w + 3]);
        }
    }
    __syncthreads();
    const float floor_ = a.relu ? 0.f : -__builtin_inff();
    const int Ho = POOL ? a.H >> 1 : a.H, Wo = POOL ? a.W >> 1 : a.W;         // output grid
    constexpr int ST = POOL ? 1 : 2;                                          // output pixels per tile and axis
    const int oy0 = POOL ? y0 >> 1 : y0, ox0 = POOL ? x0 >> 1 : x0;
    float* const ubase = a.out + (((long)b * Ho + oy0) * Wo + ox0) * a.Cout;  // uniform
    const int rows_left = Ho - oy0, cols_left = Wo - ox0 - ST * 4 * hh;
    // wave-uniform: every tile column of the region is inside the image, the byte offset of a row below it (dropped by the range check)
    // does not wrap around 32 bits, and the scalar offset ALONE (up to ST * 4 - 1 rows) stays below the record count: the raw-buffer
    // check is `offset >= num_records - soffset`, whose right side must not wrap for a map lower than one region (small tiles)
    const bool cols_all = Wo - ox0 >= ST * 8 && Ho >= ST * 4 && (unsigned long)(Ho + 8) * (unsigned long)Wo * (unsigned long)a.Cout * 4ul < (1ul << 32);
    const __amdgpu_buffer_rsrc_t rout = wmake_rsrc(a.out + (long)b * Ho * Wo * a.Cout, (unsigned)Ho * Wo * a.Cout * 4u);
    auto finish_row = [&](auto W_) {
        constexpr int w = decltype(W_)::value;                                 // this wave's tile row: registers 4 w .. 4 w + 3
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int co = co0 + n * 32 + c;
            const float bv = a.bias[co];
            const unsigned lane_off = (unsigned)(ST * 4 * hh) * a.Cout + co;
            float4 y[2][2];                                                     // [a][b], four tile columns each
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                float4 sr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    sr[r] = r == w ? make_float4(sv[bb][n][4 * w], sv[bb][n][4 * w + 1], sv[bb][n][4 * w + 2], sv[bb][n][4 * w + 3])
                                   : xw[(((r * 4 + w) * 2 + bb) * 2 + n) * 64 + lane];
                const float4 bq = make_float4(bv, bv, bv, bv);
                y[0][bb] = add4(add4(add4(sr[0], sr[1]), sr[2]), bq);
                y[1][bb] = add4(sub4(sr[1], add4(sr[2], sr[3]), m1), bq);
            }
            const float* y00 = reinterpret_cast<const float*>(&y[0][0]), *y01 = reinterpret_cast<const float*>(&y[0][1]);
            const float* y10 = reinterpret_cast<const float*>(&y[1][0]), *y11 = reinterpret_cast<const float*>(&y[1][1]);
            if constexpr (!POOL) {       // ReLU on the register pairs (on the scalars each maximum drags two canonicalising v_max along)
                const f32x2 fl2 = {floor_, floor_};
#pragma unroll
                for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) {
                        const f32x2 lo = __builtin_elementwise_max(f32x2{y[aa][bb].x, y[aa][bb].y}, fl2), hi = __builtin_elementwise_max(f32x2{y[aa][bb].z, y[aa][bb].w}, fl2);
                        y[aa][bb] = make_float4(lo.x, lo.y, hi.x, hi.y);
                    }
            }
            if (cols_all) {
                // the region's columns all lie inside the image (always, unless the width is ragged): stores through a buffer descriptor of
                // this image's output - one per-lane offset for the whole epilogue, the pixel as a SCALAR offset, rows past the image
                // dropped by the range check: no 64-bit address arithmetic, no compare, no branch per store; ReLU as one v_med3
                const unsigned lane_voff = ((unsigned)(oy0 * Wo + ox0 + ST * 4 * hh) * a.Cout + co) * 4u;
                const unsigned row_b = (unsigned)Wo * a.Cout * 4u, px_b = (unsigned)a.Cout * 4u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v00 = y00[q], v01 = y01[q], v10 = y10[q], v11 = y11[q];
                    if constexpr (POOL) {
                        const float mx = __builtin_fmaxf(__builtin_fmaxf(v00, v01), v10);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_fmaxf(__builtin_fmaxf(mx, v11), floor_)), rout, lane_voff, w * row_b + q * px_b, 0);
                    } else {
                        const unsigned so = (2 * w) * row_b + (2 * q) * px_b;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v00), rout, lane_voff, so, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v01), rout, lane_voff, so + px_b, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v10), rout, lane_voff, so + row_b, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v11), rout, lane_voff, so + row_b + px_b, 0);
                    }
                }
                continue;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {                                       // tile column q (+ 4 hh through lane_off)
                const float v00 = y00[q], v01 = y01[q], v10 = y10[q], v11 = y11[q];
                if constexpr (POOL) {      // relu(max) = max(relu)
                    float* const up = ubase + ((long)w * Wo + q) * a.Cout;
                    const float m = __builtin_fmaxf(__builtin_fmaxf(v00, v01), v10);
                    if (w < rows_left && q < cols_left) up[lane_off] = __builtin_fmaxf(__builtin_fmaxf(m, v11), floor_);
                } else {
                    float* const up = ubase + ((long)(2 * w) * Wo + 2 * q) * a.Cout;
                    float* const dn = up + (long)Wo * a.Cout;
                    const bool c0 = 2 * q < cols_left, c1 = 2 * q + 1 < cols_left;
                    if (2 * w < rows_left) {
                        if (c0) up[lane_off] = v00;
                        if (c1) up[lane_off + a.Cout] = v01;
                    }
                    if (2 * w + 1 < rows_left) {
                        if (c0) dn[lane_off] = v10;
                        if (c1) dn[lane_off + a.Cout] = v11;
                    }
                }
            }
        }
    };
    if (ph == 0) finish_row(std::integral_constant<int, 0>{});
    else if (ph == 1) finish_row(std::integral_constant<int, 1>{});
    else if (ph == 2) finish_row(std::integral_constant<int, 2>{});
    else finish_row(std::integral_constant<int, 3>{});
}

// The same epilogue (same operations per output, same bits) for the persistent kernel of conv_wino_bx2.hip: the exchange goes in two rounds
// (output channels n * 32 ..) through a 24 KB image per group that nothing else ever occupies (transfers of the next region are in flight under
// it), the bias comes in registers (no vector-memory LOAD may be issued here: the kernel counts its transfers in vmcnt), and the stores of the
// fast path are counted: returns true when the wave issued exactly WINO_ROUNDS_STORES<POOL> buffer stores and nothing else, false when it took
// the compared-store path (ragged right edge, maps lower than a region). Contains THREE __syncthreads().
template <bool POOL> static constexpr int WINO_ROUNDS_STORES = POOL ? 8 : 32;
template <bool POOL>
__device__ __forceinline__ bool wino_epilogue_rounds(const ConvArgs& a, const f32x16 (&acc)[8], float4* const xw, const int ph, const int lane, const int b,
                                                     const int y0, const int x0, const int co0, const f32x2 m1, const float (&bias2)[2]) {
    const int c = lane & 31, hh = lane >> 5;
    // ---- inverse transform Y = A^T M A. acc[2 j + n]: position (ph, j), output channels n * 32 + c. Pass over j in registers:
    //   s_ph[0] = (M0 + M1) + M2,  s_ph[1] = (M1 - M2) - M3;   then over the four V rows = the four waves:
    //   Y[0][b] = (s0[b] + s1[b]) + s2[b],   Y[1][b] = s1[b] - (s2[b] + s3[b])       (the associations of the two-row form: same bits)
    // Wave w finishes tile row w (accumulator registers 4 w .. 4 w + 3 of every s): each wave hands the other three the four registers
    // they own of its four s vectors ([source][owner][b][n][lane] float4: 12 writes, 12 reads per lane, one barrier).
    f32x16 sv[2][2];     // [b][n]
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        sv[0][n] = (acc[0 + n] + acc[2 + n]) + acc[4 + n];
        sv[1][n] = sub16(sub16(acc[2 + n], acc[4 + n], m1), acc[6 + n], m1);
    }
    const float floor_ = a.relu ? 0.f : -__builtin_inff();
    const int Ho = POOL ? a.H >> 1 : a.H, Wo = POOL ? a.W >> 1 : a.W;         // output grid
    constexpr int ST = POOL ? 1 : 2;                                          // output pixels per tile and axis
    const int oy0 = POOL ? y0 >> 1 : y0, ox0 = POOL ? x0 >> 1 : x0;
    float* const ubase = a.out + (((long)b * Ho + oy0) * Wo + ox0) * a.Cout;  // uniform
    const int rows_left = Ho - oy0, cols_left = Wo - ox0 - ST * 4 * hh;
    // wave-uniform: every tile column of the region is inside the image, the byte offset of a row below it (dropped by the range check)
    // does not wrap around 32 bits, and the scalar offset ALONE (up to ST * 4 - 1 rows) stays below the record count: the raw-buffer
    // check is `offset >= num_records - soffset`, whose right side must not wrap for a map lower than one region (small tiles)
    const bool cols_all = Wo - ox0 >= ST * 8 && Ho >= ST * 4 && (unsigned long)(Ho + 8) * (unsigned long)Wo * (unsigned long)a.Cout * 4ul < (1ul << 32);
    const __amdgpu_buffer_rsrc_t rout = wmake_rsrc(a.out + (long)b * Ho * Wo * a.Cout, (unsigned)Ho * Wo * a.Cout * 4u);
    auto finish_row = [&](auto W_) {
        constexpr int w = decltype(W_)::value;                                 // this wave's tile row: registers 4 w .. 4 w + 3
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if (n == 1) __syncthreads();                                       // every wave has read round 0
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                if (d != w) {
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb)
                        xw[((w * 3 + (d - (d > w ? 1 : 0))) * 2 + bb) * 64 + lane] =
                            make_float4(sv[bb][n][4 * d], sv[bb][n][4 * d + 1], sv[bb][n][4 * d + 2], sv[bb][n][4 * d + 3]);
                }
            }
            __syncthreads();
            const int co = co0 + n * 32 + c;
            const float bv = bias2[n];
            const unsigned lane_off = (unsigned)(ST * 4 * hh) * a.Cout + co;
            float4 y[2][2];                                                     // [a][b], four tile columns each
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                float4 sr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    sr[r] = r == w ? make_float4(sv[bb][n][4 * w], sv[bb][n][4 * w + 1], sv[bb][n][4 * w + 2], sv[bb][n][4 * w + 3])
                                   : xw[((r * 3 + (w - (w > r ? 1 : 0))) * 2 + bb) * 64 + lane];
                const float4 bq = make_float4(bv, bv, bv, bv);
                y[0][bb] = add4(add4(add4(sr[0], sr[1]), sr[2]), bq);
                y[1][bb] = add4(sub4(sr[1], add4(sr[2], sr[3]), m1), bq);
            }
            const float* y00 = reinterpret_cast<const float*>(&y[0][0]), *y01 = reinterpret_cast<const float*>(&y[0][1]);
            const float* y10 = reinterpret_cast<const float*>(&y[1][0]), *y11 = reinterpret_cast<const float*>(&y[1][1]);
            if constexpr (!POOL) {       // ReLU on the register pairs (on the scalars each maximum drags two canonicalising v_max along)
                const f32x2 fl2 = {floor_, floor_};
#pragma unroll
                for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) {
                        const f32x2 lo = __builtin_elementwise_max(f32x2{y[aa][bb].x, y[aa][bb].y}, fl2), hi = __builtin_elementwise_max(f32x2{y[aa][bb].z, y[aa][bb].w}, fl2);
                        y[aa][bb] = make_float4(lo.x, lo.y, hi.x, hi.y);
                    }
            }
            if (cols_all) {
                // the region's columns all lie inside the image (always, unless the width is ragged): stores through a buffer descriptor of
                // this image's output - one per-lane offset for the whole epilogue, the pixel as a SCALAR offset, rows past the image
                // dropped by the range check: no 64-bit address arithmetic, no compare, no branch per store; ReLU as one v_med3
                const unsigned lane_voff = ((unsigned)(oy0 * Wo + ox0 + ST * 4 * hh) * a.Cout + co) * 4u;
                const unsigned row_b = (unsigned)Wo * a.Cout * 4u, px_b = (unsigned)a.Cout * 4u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v00 = y00[q], v01 = y01[q], v10 = y10[q], v11 = y11[q];
                    if constexpr (POOL) {
                        const float mx = __builtin_fmaxf(__builtin_fmaxf(v00, v01), v10);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_fmaxf(__builtin_fmaxf(mx, v11), floor_)), rout, lane_voff, w * row_b + q * px_b, 0);
                    } else {
                        const unsigned so = (2 * w) * row_b + (2 * q) * px_b;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v00), rout, lane_voff, so, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v01), rout, lane_voff, so + px_b, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v10), rout, lane_voff, so + row_b, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v11), rout, lane_voff, so + row_b + px_b, 0);
                    }
                }
                continue;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {                                       // tile column q (+ 4 hh through lane_off)
                const float v00 = y00[q], v01 = y01[q], v10 = y10[q], v11 = y11[q];
                if constexpr (POOL) {      // relu(max) = max(relu)
                    float* const up = ubase + ((long)w * Wo + q) * a.Cout;
                    const float m = __builtin_fmaxf(__builtin_fmaxf(v00, v01), v10);
                    if (w < rows_left && q < cols_left) up[lane_off] = __builtin_fmaxf(__builtin_fmaxf(m, v11), floor_);
                } else {
                    float* const up = ubase + ((long)(2 * w) * Wo + 2 * q) * a.Cout;
                    float* const dn = up + (long)Wo * a.Cout;
                    const bool c0 = 2 * q < cols_left, c1 = 2 * q + 1 < cols_left;
                    if (2 * w < rows_left) {
                        if (c0) up[lane_off] = v00;
                        if (c1) up[lane_off + a.Cout] = v01;
                    }
                    if (2 * w + 1 < rows_left) {
                        if (c0) dn[lane_off] = v10;
                        if (c1) dn[lane_off + a.Cout] = v11;
                    }
                }
            }
        }
    };
    if (ph == 0) finish_row(std::integral_constant<int, 0>{});
    else if (ph == 1) finish_row(std::integral_constant<int, 1>{});
    else if (ph == 2) finish_row(std::integral_constant<int, 2>{});
    else finish_row(std::integral_constant<int, 3>{});
    return cols_all;
}


}  // namespace im
