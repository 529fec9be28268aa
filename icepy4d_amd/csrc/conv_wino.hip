// Winograd F(2x2, 3x3) form of the SuperPoint 3x3 convolutions (same layers as conv.hip: `lightglue/superpoint.py:155-168,
// 203`), still on the f32-input matrix cores: 16 element-wise products per 2x2 output tile instead of 36, i.e. 2.25x
// fewer MFMA FLOPs than the direct implicit GEMM for the same result in exact arithmetic (fp32 rounding differs in the
// last bits, like any change of summation order; checked against an fp64 reference in tests/test_gpu_kernels.py).
//
//   V = B^T d B   (4x4 input tile d, per channel)          B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
//   U = G g G^T   (3x3 filter g, transformed on the host)  G   = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
//   M[p] = sum_cin V[p] * U[p]   for the 16 positions p    -> 16 GEMMs  [tiles x Cin] x [Cin x Cout]
//   Y = A^T M A   (2x2 outputs)                            A^T = [1 1 1 0; 0 1 -1 -1]
//
// On gfx950 every VALU / LDS instruction a wave issues costs matrix-pipe time (tools/mfma_peak.hip: a v_fma next to a
// v_mfma_f32_32x32x2_f32 stream costs its full 4 cycles, with one or two waves per SIMD alike), so the kernel is built
// for instruction count per MFMA, not for overlap:
//   * 256 threads = 4 waves (cb, ph), output region 8 x 16 pixels = 32 Winograd tiles x 64 output channels; a wave owns
//     the 32 tiles x 32 output channels x the 8 Winograd positions of V rows 2 ph, 2 ph + 1 (128 accumulator registers);
//     TWO blocks share a CU and drift apart, so one block's prologue, barrier waits, epilogue and store drain run under
//     the other block's MFMAs;
//   * the input transform happens in REGISTERS, there is no V image: the halo patch lives in LDS as
//     [channel quad][row][column parity][column / 2] float4 (row stride 20 slots, 2 x 9 used), so the 12 input pixels a
//     lane needs for its tile and its two V rows are 12 conflict-free ds_read_b128 with compile-time offsets; the two
//     1-D passes of B^T d B are 32 packed-fp32 adds for 4 channels, and the results ARE the MFMA A operands;
//   * per 8-channel slab the U block [16][64][8] and the patch are double-buffered and filled by LDS-DMA
//     (`buffer_load_dwordx4 ... lds`): lane-linear destinations, no staging registers, no ds_write, lane offsets fixed
//     for the whole kernel, the slab a scalar offset; out-of-image pixels and padding slots read zero through the
//     descriptor's range check. One barrier per slab;
//   * epilogue: row pass of A^T M A in registers, the two V-row halves swap half of their partials through LDS (8
//     float4 per lane each way) and each finishes 16 of the 32 tiles: bias, ReLU, optional 2x2 max-pool (= one tile);
//   * FUSE1A: the patch is conv1a(img / 255) computed on the fly: thread t keeps the 3 x 3 image neighbourhood of its
//     patch pixel in registers for the whole kernel, and per slab the conv1a weights of a channel quad are wave-uniform
//     (scalar loads), so a patch value costs 9 fused multiply-adds per channel and no LDS read.
//
// History (same tests, conv 64 -> 64 + pool at 1080p, non-fused): 8-wave kernel with a V image in LDS 2.08 ms; its
// 4-wave form with a register-local inverse transform 2.08; register input transform, 8 waves, one block per CU 1.60
// (ablations: no MFMA 0.82, prologue + one slab + epilogue 0.48, no epilogue 1.39); two blocks per CU 1.53 ms; round 3: LDS-DMA
// issued as asm (no forced wait in front of the step's LDS reads) 1.49; instruction-count work found with tools/isa_census.py -
// accumulators started from a literal-zero C operand instead of 256 v_mov per wave 1.445; epilogue in vectors over neighbouring
// accumulator registers (no moves), packed subtracts (v_pk_fma_f32 by an opaque -1), uniform output addressing and max3 for
// ReLU + pool, about 500 -> 170 vector instructions per wave, 1.42 ms (fused first layer 1.49 -> 1.43). Round 4 measured three more
// ideas and adopted none (tools/experiments/conv_wino_round4_experiments.patch): the fused layer's U transfer as asm
// with its conv1a weights moved to the kernel-argument segment (keeps the scalar loads): +-0; conv1a itself on the matrix pipe
// (v_mfma_f32_4x4x1, bit-identical): +2..5 % SLOWER; an L2 prefetch of the later cache lines of every patch pixel: +5 % slower; persistent
// blocks that request the next region's first stage during the last slab (no prologue): -1 % on the large layers, +3..9 % on the small. The
// slab loop is bound by instruction issue (32 MFMAs + 20 LDS reads + ~45 vector + 10 transfer instructions per wave and step: the
// same loop without reads and transform reaches 0.95 of the matrix peak, tools/dma_rate.hip).
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

template <bool POOL, bool FUSE1A>
__global__ __launch_bounds__(256, 2) void conv3x3_wino_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sP = smem;                   // [2][S_SP]
    float* sU = smem + 2 * S_SP;        // [2][16][64][8]
    float* sX = smem;                   // epilogue exchange [2][8][128] float4, aliases the above
    float* sImg = smem + S_LDS_FLOATS;  // FUSE1A only

    const int nslices = a.Cout / 64;
    const int tx = (a.W + S_TW - 1) / S_TW, ty = (a.H + S_TH - 1) / S_TH;
    const int ntile = tx * ty * a.B;
    const int bid = blockIdx.x;
    const int rtile = (bid & 7) + 8 * ((bid >> 3) / nslices);   // XCD-aware: slices of one region share bid % 8
    const int co0 = ((bid >> 3) % nslices) * 64;
    if (rtile >= ntile) return;
    const int b = rtile / (tx * ty);
    const int trem = rtile - b * tx * ty;
    const int x0 = (trem % tx) * S_TW, y0 = (trem / tx) * S_TH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, hh = lane >> 5;
    const int cb = wave & 1, ph = wave >> 1;

    // ---- staging plan of this thread, fixed for the whole kernel: pixel slot tid of the patch image
    const int p_row = tid / S_ROW, p_rem = tid - p_row * S_ROW;
    const int p_par = p_rem >= S_PAR ? 1 : 0, p_s = p_rem - p_par * S_PAR;
    const int p_px = 2 * p_s + p_par, p_gy = y0 + p_row - 1, p_gx = x0 + p_px - 1;
    const bool p_slot = tid < S_QUAD && p_s < S_PW / 2;
    const bool p_in = p_slot && p_gy >= 0 && p_gy < a.H && p_gx >= 0 && p_gx < a.W;

    float tap[9];
    if constexpr (FUSE1A) {
        const uint8_t* img = a.img + (long)b * a.H * a.W * a.img_channels;
        for (int idx = tid; idx < S_IH * S_IW; idx += 256) {
            const int iy = idx / S_IW, ix = idx - iy * S_IW;
            const int gy = y0 + iy - 2, gx = x0 + ix - 2;
            float v = 0.f;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = image_value(img, (long)gy * a.W + gx, a.img_channels, a.gray_mode);
            sImg[idx] = v;
        }
        __syncthreads();
        const int iy = min(p_row, S_PH - 1), ix = min(p_px, S_PW - 1);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) tap[dy * 3 + dx] = sImg[(iy + dy) * S_IW + ix + dx];
    }

    const wu32x4 rin = wmake_rsrc4(FUSE1A ? (const void*)a.w : (const void*)(a.in + (long)b * a.H * a.W * a.Cin),
                                   FUSE1A ? 0u : (unsigned)a.H * a.W * a.Cin * 4u);
    const wu32x4 ruw = wmake_rsrc4(a.w, (unsigned)(a.Cin / WCC) * 16u * a.Cout * WCC * 4u);
    const unsigned lds_sP = (unsigned)(unsigned long)(lds_ptr_t)sP, lds_sU = (unsigned)(unsigned long)(lds_ptr_t)sU;   // LDS byte addresses
    // The fused first layer keeps the builtin form of the U transfer (compiler-managed waits): its stage is dominated by the conv1a
    // arithmetic and the ds_writes of the patch, and with the asm form it measured 10 % SLOWER (1.67 vs 1.52 ms at 1080p); the
    // plain layers gain 3-5 % from the asm form.
    const __amdgpu_buffer_rsrc_t ruw_b = wmake_rsrc(a.w, (unsigned)(a.Cin / WCC) * 16u * a.Cout * WCC * 4u);
    const unsigned pv = p_in ? (unsigned)(((unsigned)p_gy * a.W + p_gx) * a.Cin) * 4u : 0xFFFFF000u;
    // U image in LDS: [pos][channel quad][64 output channels] float4, so that the 16 lanes of a ds_read_b128 lane
    // group read 16 consecutive slots (with the quad innermost the even / odd slots of one quad gave a 2-way bank
    // conflict on every B-operand read). DMA slot tid + 256 k, k < 8: pos = (tid >> 7) + 2 k, quad = (tid >> 6) & 1,
    // output channel tid & 63; the source block stays [pos][Cout][8].
    const unsigned uv = (unsigned)((((tid >> 7) * a.Cout + co0 + (tid & 63)) * WCC) + ((tid >> 6) & 1) * 4) * 4u;
    const unsigned u_slab_bytes = 16u * a.Cout * WCC * 4u, u_k_bytes = 2u * a.Cout * WCC * 4u;

    auto fused_quad = [&](int ch) -> float4 {      // conv1a(img / 255) of this thread's pixel, channels ch .. ch + 3 (uniform)
        const float4 bq = *reinterpret_cast<const float4*>(a.b1 + ch);
        f32x2 lo = {bq.x, bq.y}, hi = {bq.z, bq.w};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 wv = *reinterpret_cast<const float4*>(a.w1 + t * 64 + ch);
            const f32x2 tt = {tap[t], tap[t]};
            lo = __builtin_elementwise_fma(tt, f32x2{wv.x, wv.y}, lo);
            hi = __builtin_elementwise_fma(tt, f32x2{wv.z, wv.w}, hi);
        }
        float4 v = make_float4(fmaxf(lo.x, 0.f), fmaxf(lo.y, 0.f), fmaxf(hi.x, 0.f), fmaxf(hi.y, 0.f));
        if (!p_in) v = make_float4(0.f, 0.f, 0.f, 0.f);
        return v;
    };
    // fill stage (slab & 1) with slab's U block and patch
#define IM_SSTAGE(slab)                                                                                 \
    {                                                                                                   \
        const unsigned ub = lds_sU + (((slab) & 1) * W_SU + wave * 256) * 4u;                           \
        const unsigned so_ = (slab) * u_slab_bytes;                                                     \
        _Pragma("unroll") for (int k_ = 0; k_ < 8; ++k_) {                                              \
            if constexpr (FUSE1A) __builtin_amdgcn_raw_ptr_buffer_load_lds(ruw_b, (lds_ptr_t)(sU + ((slab) & 1) * W_SU + wave * 256 + k_ * 1024), 16, uv, so_ + k_ * u_k_bytes, 0, 0); \
            else dma16(ruw, ub + k_ * 4096u, uv, so_ + k_ * u_k_bytes);                                 \
        }                                                                                               \
        float* pb = sP + ((slab) & 1) * S_SP;                                                           \
        if (tid < S_QUAD) {                                                                             \
            if constexpr (FUSE1A) {                                                                     \
                reinterpret_cast<float4*>(pb)[tid] = fused_quad((slab) * WCC);                          \
                reinterpret_cast<float4*>(pb)[S_QUAD + tid] = fused_quad((slab) * WCC + 4);             \
            } else {                                                                                    \
                const unsigned pb_ = lds_sP + (((slab) & 1) * S_SP + wave * 256) * 4u;                  \
                dma16(rin, pb_, pv, (slab) * (WCC * 4u));                                               \
                dma16(rin, pb_ + S_QUAD * 16u, pv, (slab) * (WCC * 4u) + 16u);                          \
            }                                                                                           \
        }                                                                                               \
    }

    f32x16 acc[8];   // position (2 ph + (p >> 2), p & 3); first written by the MFMAs of slab 0, which take a literal zero as C

    // lane (c, hh): tile c of the 4 x 8 tile grid, channel quad hh; B operand: output channel cb * 32 + c
    const int t_ty = c >> 3, t_tx = c & 7;
    const int a_slot = hh * S_QUAD + (2 * t_ty + ph) * S_ROW + t_tx;         // patch row 2 ty + ph, parity 0, quad hh
    const int b_slot = ((ph * 8) * 2 + hh) * 64 + cb * 32 + c;               // position 8 ph, quad hh, this lane's channel
#define IM_SD(i, j) pa[a_slot + (i) * S_ROW + ((j) & 1) * S_PAR + ((j) >> 1)]
#define IM_SMMA(slab, FIRST)                                                                            \
    {                                                                                                   \
        const float4* pa = reinterpret_cast<const float4*>(sP + ((slab) & 1) * S_SP);                   \
        const float4* ua = reinterpret_cast<const float4*>(sU + ((slab) & 1) * W_SU) + b_slot;          \
        float4 v[8];                                                                                    \
        {                                                                                               \
            float4 t0[4], t1[4];                                                                        \
            if (ph == 0) {              /* V rows 0, 1 from patch rows 0, 1, 2 */                       \
                _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                      \
                    const float4 d0 = IM_SD(0, j_), d1 = IM_SD(1, j_), d2 = IM_SD(2, j_);               \
                    t0[j_] = sub4(d0, d2, m1); t1[j_] = add4(d1, d2);                                   \
                }                                                                                       \
            } else {                    /* V rows 2, 3 from patch rows 1, 2, 3 (= rows 0, 1, 2 relative to ph) */ \
                _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                      \
                    const float4 d1 = IM_SD(0, j_), d2 = IM_SD(1, j_), d3 = IM_SD(2, j_);               \
                    t0[j_] = sub4(d2, d1, m1); t1[j_] = sub4(d1, d3, m1);                               \
                }                                                                                       \
            }                                                                                           \
            v[0] = sub4(t0[0], t0[2], m1); v[1] = add4(t0[1], t0[2]); v[2] = sub4(t0[2], t0[1], m1); v[3] = sub4(t0[1], t0[3], m1); \
            v[4] = sub4(t1[0], t1[2], m1); v[5] = add4(t1[1], t1[2]); v[6] = sub4(t1[2], t1[1], m1); v[7] = sub4(t1[1], t1[3], m1); \
        }                                                                                               \
        float4 u[8];                                                                                    \
        _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_) u[p_] = ua[p_ * 128];                          \
        _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_) acc[p_] = mfma32(v[p_].x, u[p_].x, (FIRST) ? f32x16{} : acc[p_]); \
        _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_) acc[p_] = mfma32(v[p_].y, u[p_].y, acc[p_]);   \
        _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_) acc[p_] = mfma32(v[p_].z, u[p_].z, acc[p_]);   \
        _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_) acc[p_] = mfma32(v[p_].w, u[p_].w, acc[p_]);   \
    }

    const int nslab = a.Cin / WCC;
    const f32x2 m1 = minus_one();
    IM_SSTAGE(0)
    if constexpr (!FUSE1A) IM_DMA_WAIT();
    __syncthreads();
    // one step: stage (slab + 1) & 1 was last read in step slab - 1 and its transfers stay in flight under this step's MFMAs;
    // the MFMAs stay in FRONT of the wait and the barrier; the wait covers this wave's transfers, the barrier the others'
    // (FUSE1A: the compiler's own vmcnt(0) in front of the barrier covers the builtin transfers)
#define IM_SSTEP(slab, FIRST)                                                                           \
    {                                                                                                   \
        if ((slab) + 1 < nslab) IM_SSTAGE((slab) + 1)                                                   \
        IM_SMMA(slab, FIRST)                                                                            \
        if constexpr (!FUSE1A) {                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                          \
            IM_DMA_WAIT();                                                                              \
        }                                                                                               \
        __syncthreads();                                                                                \
    }
    // slab 0 is peeled so that its first MFMAs start the accumulators from a literal zero: zeroing 128 registers ahead of the
    // loop was 256 v_mov per wave (the compiler emitted the zeroing twice), a fifth of the kernel's non-MFMA vector instructions,
    // and fp32 MFMA and VALU never co-execute on this part
    IM_SSTEP(0, true)
    for (int slab = 1; slab < nslab; ++slab) IM_SSTEP(slab, false)
#undef IM_SSTEP
#undef IM_SSTAGE
#undef IM_SD
#undef IM_SMMA

    // ---- inverse transform Y = A^T M A. Row pass (over j) in registers: s[il][b]; the column pass needs both V-row halves:
    //   Y[0][b] = s[0][b] + s[1][b] + s[2][b],   Y[1][b] = s[1][b] - s[2][b] - s[3][b]
    // ph 0 holds {s0 + s1, s1}, ph 1 holds {s2, s2 + s3}; ph 0 keeps accumulator registers 0..7, ph 1 registers 8..15, and each
    // sends the other its partials of the registers it gives away ([sender][slot][thread] float4). All of it in two-element
    // vectors over neighbouring accumulator registers (= neighbouring tiles of one row), which are register pairs already, so the
    // packed instructions need no moves; output addresses are a uniform base per register plus one per-lane offset.
    const f32x16 sa0 = (acc[0] + acc[1]) + acc[2], sa1 = sub16(sub16(acc[1], acc[2], m1), acc[3], m1);
    const f32x16 sb0 = (acc[4] + acc[5]) + acc[6], sb1 = sub16(sub16(acc[5], acc[6], m1), acc[7], m1);
    const f32x16 t0 = sa0 + sb0, t1 = sa1 + sb1;
    const int co = co0 + cb * 32 + c;
    const float bv = a.bias[co];
    const f32x2 bv2 = {bv, bv};
    const float floor_ = a.relu ? 0.f : -__builtin_inff();
    const f32x2 fl2 = {floor_, floor_};
    auto finish = [&](auto PH, const f32x16& e00, const f32x16& e01, const f32x16& e10, const f32x16& e11) {
        constexpr int P = decltype(PH)::value;
        constexpr int G = P == 0 ? 8 : 0;     // first register given away
        constexpr int K = P == 0 ? 0 : 8;     // first register kept
        float4* xs = reinterpret_cast<float4*>(sX) + (P * 8) * 128 + (tid & 127);
#define IM_SX(e, k) make_float4(e[G + 4 * (k)], e[G + 1 + 4 * (k)], e[G + 2 + 4 * (k)], e[G + 3 + 4 * (k)])
        xs[0 * 128] = IM_SX(e00, 0); xs[1 * 128] = IM_SX(e00, 1);
        xs[2 * 128] = IM_SX(e01, 0); xs[3 * 128] = IM_SX(e01, 1);
        xs[4 * 128] = IM_SX(e10, 0); xs[5 * 128] = IM_SX(e10, 1);
        xs[6 * 128] = IM_SX(e11, 0); xs[7 * 128] = IM_SX(e11, 1);
#undef IM_SX
        __syncthreads();
        const float4* xr = reinterpret_cast<const float4*>(sX) + ((P ^ 1) * 8) * 128 + (tid & 127);
        // register r of the accumulator is tile (r >> 2, (r & 3) + 4 hh) of the block's 4 x 8 tile grid (acc_row)
        const int Ho = POOL ? a.H >> 1 : a.H, Wo = POOL ? a.W >> 1 : a.W;         // output grid
        constexpr int ST = POOL ? 1 : 2;                                          // output pixels per tile and axis
        const int oy0 = POOL ? y0 >> 1 : y0, ox0 = POOL ? x0 >> 1 : x0;
        float* const ubase = a.out + (((long)b * Ho + oy0) * Wo + ox0) * a.Cout;  // uniform
        const unsigned lane_off = (unsigned)(ST * 4 * hh) * a.Cout + co;
        const int rows_left = Ho - oy0, cols_left = Wo - ox0 - ST * 4 * hh;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float4 x00 = xr[(0 + k) * 128], x01 = xr[(2 + k) * 128], x10 = xr[(4 + k) * 128], x11 = xr[(6 + k) * 128];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = K + 4 * k + 2 * h;                                  // registers r, r + 1
                const f32x2 a00 = h ? f32x2{x00.z, x00.w} : f32x2{x00.x, x00.y}, a01 = h ? f32x2{x01.z, x01.w} : f32x2{x01.x, x01.y};
                const f32x2 a10 = h ? f32x2{x10.z, x10.w} : f32x2{x10.x, x10.y}, a11 = h ? f32x2{x11.z, x11.w} : f32x2{x11.x, x11.y};
                f32x2 y00 = (f32x2{e00[r], e00[r + 1]} + a00) + bv2, y01 = (f32x2{e01[r], e01[r + 1]} + a01) + bv2;
                f32x2 y10, y11;
                if constexpr (P == 0) {        // own s1 minus the other's s2 + s3
                    y10 = __builtin_elementwise_fma(a10, m1, f32x2{e10[r], e10[r + 1]}) + bv2;
                    y11 = __builtin_elementwise_fma(a11, m1, f32x2{e11[r], e11[r + 1]}) + bv2;
                } else {                       // the other's s1 minus own s2 + s3
                    y10 = __builtin_elementwise_fma(f32x2{e10[r], e10[r + 1]}, m1, a10) + bv2;
                    y11 = __builtin_elementwise_fma(f32x2{e11[r], e11[r + 1]}, m1, a11) + bv2;
                }
                if constexpr (!POOL) {
                    y00 = __builtin_elementwise_max(y00, fl2); y01 = __builtin_elementwise_max(y01, fl2);
                    y10 = __builtin_elementwise_max(y10, fl2); y11 = __builtin_elementwise_max(y11, fl2);
                }
                const int trow = (K + 4 * k) >> 2 & 3;                            // r >> 2
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int tcol = 2 * h + e;                                   // (r + e) & 3
                    const float v00 = e ? y00.y : y00.x, v01 = e ? y01.y : y01.x, v10 = e ? y10.y : y10.x, v11 = e ? y11.y : y11.x;
                    if constexpr (POOL) {      // relu(max) = max(relu)
                        float* const up = ubase + ((long)trow * Wo + tcol) * a.Cout;
                        const float m = __builtin_fmaxf(__builtin_fmaxf(v00, v01), v10);
                        if (trow < rows_left && tcol < cols_left) up[lane_off] = __builtin_fmaxf(__builtin_fmaxf(m, v11), floor_);
                    } else {
                        float* const up = ubase + ((long)(2 * trow) * Wo + 2 * tcol) * a.Cout;
                        float* const dn = up + (long)Wo * a.Cout;
                        const bool c0 = 2 * tcol < cols_left, c1 = 2 * tcol + 1 < cols_left;
                        if (2 * trow < rows_left) {
                            if (c0) up[lane_off] = v00;
                            if (c1) up[lane_off + a.Cout] = v01;
                        }
                        if (2 * trow + 1 < rows_left) {
                            if (c0) dn[lane_off] = v10;
                            if (c1) dn[lane_off + a.Cout] = v11;
                        }
                    }
                }
            }
        }
    };
    if (ph == 0) finish(std::integral_constant<int, 0>{}, t0, t1, sb0, sb1);
    else finish(std::integral_constant<int, 1>{}, sa0, sa1, t0, t1);
}


// a.w must be the Winograd-packed weights [Cin/8][16][Cout][8] (pack_conv3x3_wino)
template <bool POOL, bool FUSE>
static hipError_t launch_wino(const ConvArgs& a, hipStream_t s) {
    const int ntile = ((a.W + S_TW - 1) / S_TW) * ((a.H + S_TH - 1) / S_TH) * a.B;
    dim3 grid(((ntile + 7) / 8) * 8 * (a.Cout / 64)), block(256);
    const size_t lds = (S_LDS_FLOATS + (FUSE ? S_FUSE : 0)) * sizeof(float);
    static size_t lds_optin[IM_MAX_DEVICES] = {0};   // per device: a process may hold contexts on several GPUs
    if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(&conv3x3_wino_kernel<POOL, FUSE>), lds, lds_optin); e != hipSuccess) return e;
    hipLaunchKernelGGL((conv3x3_wino_kernel<POOL, FUSE>), grid, block, lds, s, a);
    return hipGetLastError();
}

hipError_t launch_conv3x3_wino(const ConvArgs& a, hipStream_t s) {
    if (a.Cin < WCC || a.Cin % WCC != 0 || a.Cout % 64 != 0) return hipErrorInvalidValue;
    if (a.img) {
        if (a.Cin != 64 || !a.w1 || !a.b1) return hipErrorInvalidValue;
        return a.pool ? launch_wino<true, true>(a, s) : launch_wino<false, true>(a, s);
    }
    return a.pool ? launch_wino<true, false>(a, s) : launch_wino<false, false>(a, s);
}

}  // namespace im
