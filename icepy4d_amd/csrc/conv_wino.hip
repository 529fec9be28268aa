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
//   * 256 threads = 4 waves = the 4 ROWS of V, output region 8 x 16 pixels = 32 Winograd tiles x 64 output channels; wave ph
//     owns the 32 tiles x 64 output channels (two MFMA column tiles that share every A operand) x the 4 positions of V row ph
//     (128 accumulator registers). Round 4: until then a wave owned two V rows x 32 channels, so the two channel halves each
//     computed every V value of the block and each read three patch rows: 32 packed adds + 20 LDS reads per 32 MFMAs; now every V
//     value is computed once per block from the TWO patch rows its V row needs: 16 packed adds + 16 reads per 32 MFMAs (the loop is
//     bound by instruction issue, see below), bit-identical, -2..-5 % per layer;
//     TWO blocks share a CU and drift apart, so one block's prologue, barrier waits, epilogue and store drain run under
//     the other block's MFMAs;
//   * the input transform happens in REGISTERS, there is no V image: the halo patch lives in LDS as
//     [channel quad][row][column parity][column / 2] float4 (row stride 20 slots, 2 x 9 used), so the 8 input pixels a
//     lane needs for its tile and its V row are 8 conflict-free ds_read_b128 (two wave-uniform row bases, compile-time column
//     offsets); the row pass is one packed fma per pixel pair (dA + sgn dB with a wave-uniform sign), the column pass the usual
//     four combinations, and the results ARE the MFMA A operands;
//   * per 8-channel slab the U block [16][64][8] and the patch are double-buffered and filled by LDS-DMA
//     (`buffer_load_dwordx4 ... lds`): lane-linear destinations, no staging registers, no ds_write, lane offsets fixed
//     for the whole kernel, the slab a scalar offset; out-of-image pixels and padding slots read zero through the
//     descriptor's range check. One barrier per slab;
//   * epilogue: the pass of A^T M A over a V row's four positions in registers, then the four waves exchange through LDS (12
//     float4 per lane each way, one barrier) and wave w finishes tile row w for all 64 channels: bias, ReLU, optional 2x2
//     max-pool (= one tile), in the associations of the two-row form (same bits);
//   * UREG (the 64 -> 64 layers, round 4): wave ph reads only the four positions of its V row, so the U block goes from L2 straight
//     into registers (8 coalesced 16-byte loads per lane and slab, two register sets alternate) and not through LDS at all;
//   * FUSE1A: the patch is conv1a(img / 255). With UREG (the model's conv1b, RESIDENT form, round 4): computed ONCE per block for the
//     whole halo patch and all 64 channels as a [64 x 10] . [10 x 200] product on the matrix pipe (k = 0: bias x in-image mask, k =
//     1 + t: tap t - the fma chain of the vector form, same bits) into a resident 51 KB LDS image; the slab loop then has no producer,
//     no transfer and no barrier. Without UREG (unpooled fused form, unused by the model): thread t keeps the 3 x 3 image neighbourhood
//     of its patch pixel in registers and computes its pixel's 8 channels per slab (9 packed fmas per channel pair, weights as
//     wave-uniform scalar loads).
//
// History (same tests, conv 64 -> 64 + pool at 1080p, non-fused): 8-wave kernel with a V image in LDS 2.08 ms; its
// 4-wave form with a register-local inverse transform 2.08; register input transform, 8 waves, one block per CU 1.60
// (ablations: no MFMA 0.82, prologue + one slab + epilogue 0.48, no epilogue 1.39); two blocks per CU 1.53 ms; round 3: LDS-DMA
// issued as asm (no forced wait in front of the step's LDS reads) 1.49; instruction-count work found with tools/isa_census.py -
// accumulators started from a literal-zero C operand instead of 256 v_mov per wave 1.445; epilogue in vectors over neighbouring
// accumulator registers (no moves), packed subtracts (v_pk_fma_f32 by an opaque -1), uniform output addressing and max3 for
// ReLU + pool, about 500 -> 170 vector instructions per wave, 1.42 ms (fused first layer 1.49 -> 1.43). Round 4: one V row x 64
// channels per wave instead of two V rows x 32 (above): conv 64 -> 64 + pool at 540p 0.332 -> 0.319 ms, fused first layer 1.437 ->
// 1.398; U through registers 0.319 -> 0.310 / 1.398 -> 1.356; resident patch 1.356 -> 1.293. Round 4 also measured six more ideas and
// adopted none (tools/experiments/conv_wino_round4_experiments.patch): the fused layer's U transfer as asm
// with its conv1a weights moved to the kernel-argument segment (keeps the scalar loads): +-0; conv1a itself on the matrix pipe
// (v_mfma_f32_4x4x1, bit-identical): +2..5 % SLOWER; an L2 prefetch of the later cache lines of every patch pixel: +5 % slower; persistent
// blocks that request the next region's first stage during the last slab (no prologue): -1 % on the large layers, +3..9 % on the small. The
// slab loop is bound by instruction issue (32 MFMAs + 20 LDS reads + ~45 vector + 10 transfer instructions per wave and step: the
// same loop without reads and transform reaches 0.95 of the matrix peak, tools/dma_rate.hip).
#include "conv_wino.h"

namespace im {

template <bool POOL, bool FUSE1A, bool UREG, bool BX = false>
__global__ __launch_bounds__(256, 2) void conv3x3_wino_kernel(ConvArgs a) {
    static_assert(!BX || UREG, "BX reads its U fragments straight into registers");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sP = smem;                   // [2][S_SP]
    float* sU = smem + 2 * S_SP;        // [2][16][64][8]
    // RESIDENT (the fused first layer with U in registers): conv1a of the whole halo patch and all 64 channels is computed ONCE, on
    // the matrix pipe, into a resident LDS image [16 quads][S_QUAD] float4 (51 KB); the slab loop then has no producer, no transfer
    // and NO BARRIER - patch read-only, U per wave in registers. The exchange buffer of the epilogue (64 KB) aliases the patch.
    constexpr bool RESIDENT = FUSE1A && UREG;
    float* sImg = smem + (RESIDENT || BX ? 16384 : S_LDS_FLOATS);  // FUSE1A only

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
    const int ph = wave;                  // this wave's V row (0..3): 32 tiles x 64 output channels x the 4 positions of that row

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
        if constexpr (RESIDENT) {
            // conv1a as D[channel][pixel] = sum_k W[channel][k] X[k][pixel], k = 0: bias x in-image mask, k = 1 + t: tap t - the fma chain of
            // the vector form in the same order (same bits). A = weights (lane: channel c of the row tile, k = 2 s + hh), B = the patch
            // slot's image taps; a lane ends with its slot's channels (r & 3) + 8 (r >> 2) + 4 hh = one channel quad per register group,
            // i.e. the float4 the patch image holds. Seven column tiles of 32 slots cover the 200 slots; wave w takes tiles w, w + 4.
            const __amdgpu_buffer_rsrc_t rwq = wmake_rsrc(a.w1q, 64u * 2u * 8u * 4u);
            float wA[2][5];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const unsigned off = (unsigned)(((m * 32 + c) * 2 + hh) * 8) * 4u;
                const float4 w4 = gbuf_load4(rwq, off, 0u), w5 = gbuf_load4(rwq, off + 16u, 0u);
                wA[m][0] = w4.x; wA[m][1] = w4.y; wA[m][2] = w4.z; wA[m][3] = w4.w; wA[m][4] = w5.x;
            }
            float4* const sPq = reinterpret_cast<float4*>(smem);
            for (int t = wave; t < 7; t += 4) {
                const int slot = t * 32 + c;
                const int r_ = slot / S_ROW, rem_ = slot - r_ * S_ROW;
                const int par_ = rem_ >= S_PAR ? 1 : 0, s_ = rem_ - par_ * S_PAR, px_ = 2 * s_ + par_;
                const int gy = y0 + r_ - 1, gx = x0 + px_ - 1;
                const bool in = slot < S_QUAD && s_ < S_PW / 2 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                const float* ip = sImg + min(r_, S_PH - 1) * S_IW + min(px_, S_PW - 1);
                float bB[5];
#pragma unroll
                for (int k2 = 0; k2 < 5; ++k2) {
                    // hh = 0: k = 2 k2 -> mask (k2 = 0) or tap 2 k2 - 1; hh = 1: k = 2 k2 + 1 -> tap 2 k2
                    const int t0 = 2 * k2 - 1, t1 = 2 * k2;
                    const int o0 = k2 == 0 ? 0 : (t0 / 3) * S_IW + t0 % 3, o1 = (t1 / 3) * S_IW + t1 % 3;
                    float v = ip[hh ? o1 : o0];
                    if (k2 == 0 && hh == 0) v = 1.f;
                    bB[k2] = in ? v : 0.f;
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    f32x16 d = {};
#pragma unroll
                    for (int k2 = 0; k2 < 5; ++k2) d = mfma32(wA[m][k2], bB[k2], d);
                    if (slot < S_QUAD) {
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            sPq[(m * 8 + 2 * g + hh) * S_QUAD + slot] = make_float4(fmaxf(d[4 * g], 0.f), fmaxf(d[4 * g + 1], 0.f),
                                                                                   fmaxf(d[4 * g + 2], 0.f), fmaxf(d[4 * g + 3], 0.f));
                    }
                }
            }
            __syncthreads();
        } else {
            const int iy = min(p_row, S_PH - 1), ix = min(p_px, S_PW - 1);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) tap[dy * 3 + dx] = sImg[(iy + dy) * S_IW + ix + dx];
        }
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
        _Pragma("unroll") for (int k_ = 0; k_ < (UREG ? 0 : 8); ++k_) {                                 \
            if constexpr (FUSE1A) __builtin_amdgcn_raw_ptr_buffer_load_lds(ruw_b, (lds_ptr_t)(sU + ((slab) & 1) * W_SU + wave * 256 + k_ * 1024), 16, uv, so_ + k_ * u_k_bytes, 0, 0); \
            else dma16(ruw, ub + k_ * 4096u, uv, so_ + k_ * u_k_bytes);                                 \
        }                                                                                               \
        float* pb = sP + ((slab) & 1) * S_SP;                                                           \
        if (!RESIDENT && tid < S_QUAD) {                                                                \
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

    // lane (c, hh): tile c of the 4 x 8 tile grid, channel quad hh; B operands: output channels c and 32 + c.
    // V row ph of B^T d B needs TWO patch rows (0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3): 8 operand reads and 16 packed adds per
    // slab, and every V value of the block is computed exactly once (the two-row form computes each one in both channel halves).
    const int t_ty = c >> 3, t_tx = c & 7;
    const int rowA = ph == 0 ? 0 : (ph == 2 ? 2 : 1), rowB = ph == 0 ? 2 : (ph == 3 ? 3 : (ph == 2 ? 1 : 2));   // wave-uniform
    const int a_slotA = hh * S_QUAD + (2 * t_ty + rowA) * S_ROW + t_tx, a_slotB = hh * S_QUAD + (2 * t_ty + rowB) * S_ROW + t_tx;
    const int b_slot = ((ph * 4) * 2 + hh) * 64 + c;                         // position 4 ph, quad hh, this lane's channel
    f32x2 rsgn;                                                               // +1 for V row 1 (d1 + d2), -1 otherwise: dA + rsgn dB
    {
        float sg = __uint_as_float(__builtin_amdgcn_readfirstlane(ph == 1 ? 0x3F800000u : 0xBF800000u));
        asm("" : "+s"(sg));
        rsgn = f32x2{sg, sg};
    }
    // UREG (the 64 -> 64 layers): U straight from L2 into registers. Wave ph reads only the four positions of its V row and nobody
    // else reads them, so the U block need not pass through LDS at all: per slab 8 coalesced 16-byte loads per lane (32 output
    // channels x 32 bytes = 1 KB per load and half wave), requested one slab ahead into the other of two register sets, instead of
    // 8 LDS-DMA pieces + 8 ds_read_b128. Measured -2..-3 % at 64 output channels and +1..+3 % at 128 / 256 (several slices per
    // region fetch from L2 at once), hence a template parameter chosen per layer.
    const __amdgpu_buffer_rsrc_t ruq = wmake_rsrc(a.w, (unsigned)(a.Cin / WCC) * 16u * a.Cout * WCC * 4u);
    const unsigned uq_voff = (unsigned)((((ph * 4) * a.Cout + co0 + c) * WCC) + hh * 4) * 4u;
    const unsigned uq_j = (unsigned)a.Cout * WCC * 4u;
    float4 uA[8], uB[8];
#define IM_ULOAD(dst, slab) _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_) dst[p_] = gbuf_load4(ruq, uq_voff + (p_ & 1) * 1024u, (unsigned)(slab) * u_slab_bytes + (p_ >> 1) * uq_j);
#define IM_SDA(j) pa[a_slotA + ((j) & 1) * S_PAR + ((j) >> 1)]
#define IM_SDB(j) pa[a_slotB + ((j) & 1) * S_PAR + ((j) >> 1)]
#define IM_SMMA(slab, FIRST) IM_SMMA_U(slab, FIRST, ul_)
#define IM_SMMA_U(slab, FIRST, u)                                                                       \
    {                                                                                                   \
        const float4* pa = RESIDENT ? reinterpret_cast<const float4*>(smem) + (slab) * (2 * S_QUAD)    \
                                    : reinterpret_cast<const float4*>(sP + ((slab) & 1) * S_SP);        \
        const float4* ua = reinterpret_cast<const float4*>(sU + ((slab) & 1) * W_SU) + b_slot;          \
        float4 v[4];                                                                                    \
        {                                                                                               \
            float4 t[4];                                                                                \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) t[j_] = sub4(IM_SDA(j_), IM_SDB(j_), rsgn); \
            v[0] = sub4(t[0], t[2], m1); v[1] = add4(t[1], t[2]); v[2] = sub4(t[2], t[1], m1); v[3] = sub4(t[1], t[3], m1); \
        }                                                                                               \
        float4 ul_[8];                                                                                  \
        if constexpr (!UREG) { _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_) ul_[p_] = ua[(p_ >> 1) * 128 + (p_ & 1) * 32]; } \
        _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_) acc[p_] = mfma32(v[p_ >> 1].x, u[p_].x, (FIRST) ? f32x16{} : acc[p_]); \
        _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_) acc[p_] = mfma32(v[p_ >> 1].y, u[p_].y, acc[p_]); \
        _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_) acc[p_] = mfma32(v[p_ >> 1].z, u[p_].z, acc[p_]); \
        _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_) acc[p_] = mfma32(v[p_ >> 1].w, u[p_].w, acc[p_]); \
    }

    const int nslab = a.Cin / WCC;
    const f32x2 m1 = minus_one();
    if constexpr (BX) {
        // ---- BX main loop: 16-channel chunks. Per chunk and wave: 16 patch reads, the transform of its V row (64 packed-pair adds), four cuts
        // (one per position: 8 values -> 12 registers of planes), 8 steps of six MFMAs (position j = s >> 1, output channels 32 (s & 1) ..), and
        // 8 x 3 fragment loads (1 KB each, lane-linear) requested X_AHEAD steps ahead into a ring of X_RING register slots. The patch of the
        // next chunk travels by LDS-DMA under this chunk's products (plain layers; the fused first layer reads its resident image).
        const int nchunk = a.Cin / 16;
        const unsigned ux_pos = (unsigned)(a.Cout / 32) * 3072u, ux_chunk = 16u * ux_pos;
        const __amdgpu_buffer_rsrc_t rux = wmake_rsrc(a.wx, (unsigned)nchunk * ux_chunk);
#ifdef IM_XABL_NO_U
        const unsigned ux_voff = 0x80000000u + (unsigned)lane * 16u;
#else
        const unsigned ux_voff = (unsigned)lane * 16u;
#endif
        const unsigned ux_base = (unsigned)(4 * ph) * ux_pos + (unsigned)(co0 / 32) * 3072u;
        // resident image (fused first layer): [quad][row][parity][column / 2] as in the f32 form; staged chunks (plain layers): pixel-major, see XP_PITCH
        const int x_slotA = 2 * hh * S_QUAD + (2 * t_ty + rowA) * S_ROW + t_tx, x_slotB = 2 * hh * S_QUAD + (2 * t_ty + rowB) * S_ROW + t_tx;
        const int xp_rowA = 2 * t_ty + rowA, xp_rowB = 2 * t_ty + rowB;
        const int xp_A = 2 * xp_rowA * XP_PITCH + ((xp_rowA >> 1) & 3) + t_tx * 4 + 2 * hh, xp_B = 2 * xp_rowB * XP_PITCH + ((xp_rowB >> 1) & 3) + t_tx * 4 + 2 * hh;
        wu32x4 ur[X_RING][3];
#define IM_XULOAD(slot, cbase, nbase, nvoff, s)   /* cbase / nbase: byte offsets of this chunk's and the next chunk's U planes; nvoff: the lane offset for the next chunk's */ \
        {                                                                                                \
            const unsigned so_ = IM_XABL_U_OFFSET((((s) >> 3) ? (nbase) : (cbase)) + ux_base + (unsigned)(((s) >> 1) & 3) * ux_pos + (unsigned)((s) & 1) * 3072u); \
            const unsigned vo_ = ((s) >> 3) ? (nvoff) : ux_voff;                                         \
            _Pragma("unroll") for (int pl_ = 0; pl_ < 3; ++pl_)                                            \
                ur[slot][pl_] = __builtin_amdgcn_raw_buffer_load_b128(rux, vo_ + pl_ * 1024u, so_, 0);   \
        }
        // IM_XROT (experiment): the blocks walk the input-channel chunks in rotated orders (by tile position), so that the CUs of an XCD do not all
        // ask its L2 for the same few KB of U at the same time
#ifdef IM_XROT
        const int rot = rtile % nchunk;
#else
        const int rot = 0;
#endif
        auto cc_of = [&](int it) { const int c_ = it + rot; return c_ >= nchunk ? c_ - nchunk : c_; };
        // the patch of iteration it's chunk into stage it & 1: 20 pieces (10 patch rows x 2 column parities), five per wave; lane 4 s + quad of piece
        // (row, parity) reads quad `quad` of pixel (row, 2 s + parity) - zeros outside the image through the range check; lanes past the nine used
        // slots stay out (EXEC)
        unsigned xp_voff[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int pc = wave * 5 + i, row = pc >> 1, par = pc & 1, sl = lane >> 2, quad = lane & 3;
            const int gy = y0 + row - 1, gx = x0 + 2 * sl + par - 1;
            const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            xp_voff[i] = in ? (unsigned)(((unsigned)gy * a.W + gx) * a.Cin + quad * 4) * 4u : 0xFFFFF000u;
        }
        auto xstage = [&](int it) {
            const int chunk = cc_of(it);
#ifdef IM_XABL_NO_PATCH
            if (lane > 12345)
#else
            if (lane < 4 * (S_PW / 2))
#endif
            {
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const int pc = wave * 5 + i, row = pc >> 1;
                    dma16(rin, lds_sP + (unsigned)(((it & 1) * XP_STAGE + pc * XP_PITCH + ((row >> 1) & 3)) * 16), xp_voff[i], chunk * 64u);
                }
            }
        };
        auto xchunk = [&](int it, auto FIRST_) {
            constexpr bool FIRST = decltype(FIRST_)::value;
            const bool more = it + 1 < nchunk;                                   // uniform
            const int chunk = cc_of(it);
            // past the last chunk the requests go out of range through the LANE offset (>= the descriptor's record count whatever the scalar offset
            // is: zeros, no traffic; the hardware's check is `lane offset >= records - scalar offset`, whose right side must not wrap)
            const unsigned cbase = (unsigned)chunk * ux_chunk, nbase = more ? (unsigned)cc_of(it + 1) * ux_chunk : 0u;
            const unsigned nvoff = more ? ux_voff : 0x80000000u;
            if constexpr (!RESIDENT) { if (more) xstage(it + 1); }
            const float4* pa = RESIDENT ? reinterpret_cast<const float4*>(smem) + chunk * (4 * S_QUAD)
                                        : reinterpret_cast<const float4*>(sP + (it & 1) * X_SP);
            float v[4][8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float4* pq = pa + q * S_QUAD;
                float4 t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (RESIDENT) t[j] = sub4(pq[x_slotA + (j & 1) * S_PAR + (j >> 1)], pq[x_slotB + (j & 1) * S_PAR + (j >> 1)], rsgn);
                    else t[j] = sub4(pa[xp_A + (j & 1) * XP_PITCH + (j >> 1) * 4 + q], pa[xp_B + (j & 1) * XP_PITCH + (j >> 1) * 4 + q], rsgn);
                }
                const float4 w0 = sub4(t[0], t[2], m1), w1 = add4(t[1], t[2]), w2 = sub4(t[2], t[1], m1), w3 = sub4(t[1], t[3], m1);
                v[0][4 * q] = w0.x; v[0][4 * q + 1] = w0.y; v[0][4 * q + 2] = w0.z; v[0][4 * q + 3] = w0.w;
                v[1][4 * q] = w1.x; v[1][4 * q + 1] = w1.y; v[1][4 * q + 2] = w1.z; v[1][4 * q + 3] = w1.w;
                v[2][4 * q] = w2.x; v[2][4 * q + 1] = w2.y; v[2][4 * q + 2] = w2.z; v[2][4 * q + 3] = w2.w;
                v[3][4 * q] = w3.x; v[3][4 * q + 1] = w3.y; v[3][4 * q + 2] = w3.z; v[3][4 * q + 3] = w3.w;
            }
            // the cut of position j + 1 rides in the two steps of position j (pairs 0, 1 beside the first 32 output channels' MFMAs, pairs 2, 3 beside
            // the second's); a scheduling fence per step keeps every step's fragment loads X_AHEAD steps in front of their use (the machine
            // scheduler otherwise sinks each load to its use, wait and all)
            unsigned ph_[2][4], pm_[2][4], pl_[2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) wsplit2(v[0][2 * i], v[0][2 * i + 1], ph_[0][i], pm_[0][i], pl_[0][i]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int j = s >> 1, n = s & 1, cur = j & 1;
                const wu32x4 ah = {ph_[cur][0], ph_[cur][1], ph_[cur][2], ph_[cur][3]}, am = {pm_[cur][0], pm_[cur][1], pm_[cur][2], pm_[cur][3]},
                             al = {pl_[cur][0], pl_[cur][1], pl_[cur][2], pl_[cur][3]};
                const wu32x4 bh = ur[s % X_RING][0], bm = ur[s % X_RING][1], bl = ur[s % X_RING][2];
                f32x16 x = mfma_bx(ah, bl, FIRST ? f32x16{} : acc[s]);
                x = mfma_bx(al, bh, x);
                x = mfma_bx(am, bm, x);
                x = mfma_bx(ah, bm, x);
                x = mfma_bx(am, bh, x);
                acc[s] = mfma_bx(ah, bh, x);
                if (j < 3) {
#pragma unroll
                    for (int i = 2 * n; i < 2 * n + 2; ++i) wsplit2(v[j + 1][2 * i], v[j + 1][2 * i + 1], ph_[cur ^ 1][i], pm_[cur ^ 1][i], pl_[cur ^ 1][i]);
                }
                IM_XULOAD((s + X_AHEAD) % X_RING, cbase, nbase, nvoff, s + X_AHEAD)
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!RESIDENT) {
                __builtin_amdgcn_sched_barrier(0);
                // the transfers of the next patch were issued before this chunk's 24 fragment loads: "at most the X_AHEAD fragments in flight" means
                // they have landed (loads and transfers complete in issue order); the barrier covers the other waves' transfers and reads
                if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * X_AHEAD) : "memory");
                __syncthreads();
            }
        };
#pragma unroll
        for (int s = 0; s < X_AHEAD; ++s) IM_XULOAD(s, (unsigned)cc_of(0) * ux_chunk, 0u, ux_voff, s)
        if constexpr (!RESIDENT) {
            xstage(0);
            IM_DMA_WAIT();
            __syncthreads();
        }
        unsigned long long ck0 = 0, cr0 = 0;     // im_debug_clock_probe: the shader clock inside the chunk loop (AttnArgs::clock)
        if (a.clock) { ck0 = __builtin_amdgcn_s_memtime(); cr0 = __builtin_amdgcn_s_memrealtime(); }
        xchunk(0, std::true_type{});
        for (int chunk = 1; chunk < nchunk; ++chunk) xchunk(chunk, std::false_type{});
        if (a.clock) {
            const unsigned long long ck1 = __builtin_amdgcn_s_memtime(), cr1 = __builtin_amdgcn_s_memrealtime();
            if (tid == 0) {
                unsigned long long* cp = a.clock + 2 * (blockIdx.x % CLOCK_PROBE_SLOTS);
                cp[0] = ck1 - ck0; cp[1] = cr1 - cr0;
            }
        }
#undef IM_XULOAD
    } else {
    if constexpr (UREG) { IM_ULOAD(uA, 0) }
    if constexpr (!RESIDENT) {
        IM_SSTAGE(0)
        if constexpr (!FUSE1A) IM_DMA_WAIT();
        __syncthreads();
    }
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
#define IM_USTEP(slab, FIRST, ucur, unext, LAST)                                                        \
    {                                                                                                   \
        if (!(LAST)) {                                                                                  \
            if constexpr (!RESIDENT) IM_SSTAGE((slab) + 1)                                              \
            IM_ULOAD(unext, (slab) + 1)                                                                 \
        }                                                                                               \
        IM_SMMA_U(slab, FIRST, ucur)                                                                    \
        if constexpr (!FUSE1A && !RESIDENT) {                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                          \
            if (LAST) IM_DMA_WAIT(); else IM_DMA_WAIT_BEFORE_8_LOADS();                                 \
        }                                                                                               \
        if constexpr (!RESIDENT) __syncthreads();      /* resident patch + U in registers: nothing to wait for */ \
    }
    if constexpr (UREG) {      // two register sets of U alternate: steps in pairs (nslab is even)
        IM_USTEP(0, true, uA, uB, false)
        for (int slab = 1; slab + 1 < nslab; slab += 2) {
            IM_USTEP(slab, false, uB, uA, false)
            IM_USTEP(slab + 1, false, uA, uB, false)
        }
        IM_USTEP(nslab - 1, false, uB, uA, true)
    } else {
        IM_SSTEP(0, true)
        for (int slab = 1; slab < nslab; ++slab) IM_SSTEP(slab, false)
    }
    }   // !BX
#undef IM_USTEP
#undef IM_SSTEP
#undef IM_SSTAGE
#undef IM_SDA
#undef IM_SDB
#undef IM_SMMA

#ifdef IM_XABL_NO_EPI
    if constexpr (BX) {
        float keep_ = 0.f;
#pragma unroll
        for (int p_ = 0; p_ < 8; ++p_) keep_ += acc[p_][0] + acc[p_][15];
        if (keep_ == 12345.678f) a.out[0] = keep_;
        return;
    }
#endif
    if constexpr (RESIDENT) __syncthreads();                 // the exchange buffer aliases the resident patch: every wave must be past its last slab
    // both stages are free now: the exchange image is 4 x 4 x 4 x 64 float4 = 64 KB from the start of the block's LDS
    wino_epilogue<POOL>(a, acc, reinterpret_cast<float4*>(smem), ph, lane, b, y0, x0, co0, m1);
}


// a.w must be the Winograd-packed weights [Cin/8][16][Cout][8] (pack_conv3x3_wino); BX: a.wx the bf16 planes (pack_conv3x3_wino_bx)
template <bool POOL, bool FUSE, bool UREG, bool BX = false>
static hipError_t launch_wino(const ConvArgs& a, hipStream_t s) {
    const int ntile = ((a.W + S_TW - 1) / S_TW) * ((a.H + S_TH - 1) / S_TH) * a.B;
    dim3 grid(((ntile + 7) / 8) * 8 * (a.Cout / 64)), block(256);
    const size_t lds = ((BX ? X_LDS_FLOATS : FUSE && UREG ? 16384 : S_LDS_FLOATS) + (FUSE ? S_FUSE : 0)) * sizeof(float);
    static size_t lds_optin[IM_MAX_DEVICES] = {0};   // per device: a process may hold contexts on several GPUs
    if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(&conv3x3_wino_kernel<POOL, FUSE, UREG, BX>), lds, lds_optin); e != hipSuccess) return e;
    hipLaunchKernelGGL((conv3x3_wino_kernel<POOL, FUSE, UREG, BX>), grid, block, lds, s, a);
    return hipGetLastError();
}

hipError_t launch_conv3x3_wino(const ConvArgs& a, hipStream_t s) {
    if (a.Cin < WCC || a.Cin % WCC != 0 || a.Cout % 64 != 0) return hipErrorInvalidValue;
    // BX (round 6): the products on the bf16 matrix cores whenever the bf16 planes of U are there; IM_CONV_F32=1 (read per call: the parity tests
    // run both forms in one process) or a.f32_form keeps the f32-input form of rounds 2-5, which needs a.w
    const char* const f32_env = getenv("IM_CONV_F32");
    const bool bx = a.wx && a.Cin % 16 == 0 && !(a.f32_form || (f32_env && f32_env[0] == '1'));
    if (bx) {
        const char* const bx2_env = getenv("IM_CONV_BX2");
        if (!a.img && bx2_env && bx2_env[0] == '1') return launch_conv3x3_wino_bx2(a, s);
        if (a.img) {
            if (a.Cin != 64 || !a.pool || !a.w1 || !a.b1 || !a.w1q) return hipErrorInvalidValue;     // the model's conv1b: fused first layer, pooled
            return launch_wino<true, true, true, true>(a, s);
        }
        return a.pool ? launch_wino<true, false, true, true>(a, s) : launch_wino<false, false, true, true>(a, s);
    }
    if (!a.w) return hipErrorInvalidValue;
    // U through registers where one slice covers all output channels (the 64 -> 64 layers), through LDS otherwise; the register form
    // walks the slabs in pairs
    const bool ureg = a.Cout == 64 && (a.Cin / WCC) % 2 == 0;
    if (a.img) {
        if (a.Cin != 64 || !a.w1 || !a.b1 || !a.w1q) return hipErrorInvalidValue;
        if (ureg && a.pool) return launch_wino<true, true, true>(a, s);     // (the unpooled fused form would spill with U in registers: LDS form)
        return a.pool ? launch_wino<true, true, false>(a, s) : launch_wino<false, true, false>(a, s);
    }
    if (ureg) return a.pool ? launch_wino<true, false, true>(a, s) : launch_wino<false, false, true>(a, s);
    return a.pool ? launch_wino<true, false, false>(a, s) : launch_wino<false, false, false>(a, s);
}

}  // namespace im
