// BX2 (round 6): the Winograd F(2x2, 3x3) layer on the bf16 matrix cores (six bf16 products per fp32 product, as conv_wino.hip BX - same
// arithmetic per output, same bits) with the U planes SHARED by two tile groups of one block and every byte moved by LDS-DMA.
//
// Why: in conv_wino.hip BX every wave loads its own U fragments into registers - 393 KB per 32 tiles x 64 output channels at 64 input
// channels, i.e. 64 bytes per clock and CU at the full MFMA rate, the whole L1 -> register path (`profiles/r06_conv_bx_ablations.txt`: the
// same loop is as slow with every fragment load hitting L1, and its patch transfers are waited for by the in-order vmcnt of the fragment
// loads three steps after they were issued). Here
//   * a block = 512 threads = 8 waves = 2 tile groups (g = wave >> 2: 8 x 16 output pixels = 32 tiles each, one 16 x 16 region together) x the
//     4 V rows (ph = wave & 3), one block per CU; wave (g, ph) owns 32 tiles x 64 output channels x the 4 positions of V row ph (128
//     accumulators), exactly the layout and the epilogue of conv_wino.hip;
//   * the blocks are PERSISTENT (one per CU; block b works on the regions b, b + grid, ...: always the same 64 output channels, so its U stream
//     is one endless cycle over the input-channel chunks) and the transfers run across region boundaries: the first patch and the first cuts of
//     the next region ride under the last chunk of this one. A non-persistent first version spent a third of a block's 38,000 cycles in its
//     start-up transient (`profiles/r06_conv_bx2_timeline.txt`);
//   * U travels ONCE per block: per quarter chunk (position step j of a 16-channel chunk: 4 V rows x 2 column tiles x 3 planes = 24 KB) by 24
//     LDS-DMA pieces (3 per wave) into a ring of Y_NQ quarter slots, requested Y_NQ - 1 quarters ahead; both groups read their fragments from
//     it with ds_read_b128 (lane-linear, conflict-free) one step before use: 32 bytes per clock and CU at the full MFMA rate;
//   * the 18 x 18 halo patch of a chunk is ONE stage (read in quarter 2 of the previous chunk, free again one barrier later, refilled by 3
//     pieces per wave over the next three quarters);
//   * no wave loads anything into registers from global memory in the loop, so the only vmcnt waits are the counted ones in front of the
//     quarter barriers (all transfers complete in issue order; the pattern 6 / 3 / 3 / 6 is derived below and every wave issues the same number
//     of pieces per quarter, out-of-range ones past the end, so that it holds to the last quarter; the epilogue's stores are counted in);
//   * the input transform of chunk c + 1 and its first cut ride in quarters 2 and 3 of chunk c, the cuts of positions 1..3 in quarters 0..2:
//     no wave does vector work without MFMAs in flight except in the block's prologue.
#include "conv_wino.h"

namespace im {

static constexpr int Y_TH = 16, Y_TW = 16;                 // output pixels per region: two groups of 8 rows
static constexpr int Y_PH = Y_TH + 2, Y_PW = Y_TW + 2;     // halo patch
static constexpr int Y_PIECES = 2 * Y_PH;                  // patch pieces per chunk: one per (patch row, column parity), pixel-major as conv_wino.h XP_PITCH
static constexpr int Y_PPW = (Y_PIECES + 7) / 8;           // pieces per wave and chunk (5; the 4 past the end are dummies: every wave issues the same number)
static constexpr int Y_STAGE_BYTES = (8 * Y_PPW + 1) * XP_PITCH * 16;   // 28,864: 36 pieces, 4 dummy targets, room for the row-pair shift
static constexpr int Y_NQ = 3;                             // quarter slots of the U ring
static constexpr int Y_QBYTES = 24 * 1024;                 // 4 V rows x 2 column tiles x 3 planes x 1 KB
static constexpr int Y_XBYTES = 24 * 1024;                 // exchange image of one group and one round of the epilogue
static constexpr int Y_LDS_BYTES = Y_STAGE_BYTES + Y_NQ * Y_QBYTES + 2 * Y_XBYTES;   // 151,744: stage | ring | exchange, nothing aliased (+ 256 bytes of bias)
static_assert(Y_LDS_BYTES + 256 <= 160 * 1024, "LDS plan");

#ifdef IM_YSTAMP   // diagnostic build only (tools/conv_bx2_stamps.py): shader-clock stamps of wave 0 of the first blocks, never read by the kernel
__device__ unsigned long long g_ystamp[256 * 256];
#define Y_STAMP(i) { if (wave == 0 && (i) < 256) { const unsigned long long ts_ = __builtin_amdgcn_s_memtime(); if (lane == 0) g_ystamp[blockIdx.x * 256 + (i)] = ts_; } }
#else
#define Y_STAMP(i)
#endif

struct YItem {            // one region of one image: what changes from item to item of a persistent block
    int b, y0, x0;
    unsigned pv[2];       // lane offset of a patch piece by column parity: quad `lane & 3` of pixel column 2 (lane >> 2) + parity of the patch, row 0
                          // (out of range for columns outside the image and padding slots); the piece's row rides in the scalar offset
};

template <bool POOL>
__global__ __launch_bounds__(512, 2) void conv3x3_wino_bx2_kernel(ConvArgs a, int nitems) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nslices = a.Cout / 64;
    const int tx = (a.W + Y_TW - 1) / Y_TW, ty = (a.H + Y_TH - 1) / Y_TH;
    const int ntile = tx * ty * a.B;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, ph = wave & 3;
    const int c = lane & 31, hh = lane >> 5;
    // item i: region (i & 7) + 8 ((i >> 3) / nslices), output channels 64 ((i >> 3) % nslices): the slices of one region share i % 8 (one XCD under
    // round-robin placement), and with a grid that is a multiple of 8 nslices a block keeps its slice and its i % 8 for all its items
    const int co0 = (((int)blockIdx.x >> 3) % nslices) * 64;
    auto region_of = [&](int item) __attribute__((always_inline)) { return (item & 7) + 8 * ((item >> 3) / nslices); };
    int item = blockIdx.x;
    if (item >= nitems || region_of(item) >= ntile) return;

    const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)smem;          // the patch stage
    const unsigned ldsR = lds0 + Y_STAGE_BYTES;                              // the U ring
    float4* const xw = reinterpret_cast<float4*>(reinterpret_cast<char*>(smem) + Y_STAGE_BYTES + Y_NQ * Y_QBYTES + g * Y_XBYTES);

    auto setup = [&](int it, YItem& z) __attribute__((always_inline)) {
        const int rt = region_of(it);
        z.b = rt / (tx * ty);
        const int trem = rt - z.b * tx * ty;
        z.x0 = (trem % tx) * Y_TW; z.y0 = (trem / tx) * Y_TH;
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int sl = lane >> 2, quad = lane & 3, gx = z.x0 + 2 * sl + par - 1;
            const bool in = sl < Y_PW / 2 && gx >= 0 && gx < a.W;
            z.pv[par] = in ? (unsigned)(gx * a.Cin + quad * 4) * 4u : 0x80000000u;    // out of range: zeros
        }
    };
    const int nchunk = a.Cin / 16;
    // Y_PPW pieces per wave, ALWAYS (the pieces count in vmcnt): `live` false = out of range through the lane offset (zeros, no traffic). Lanes
    // past the nine used slots of a row stay out (EXEC): their 16 bytes would land in the next piece's image
    auto patch_dma = [&](const YItem& z, int chunk, bool live) __attribute__((always_inline)) {
        const unsigned so = (live ? (unsigned)chunk : 0u) * 64u;
        const wu32x4 rs = wmake_rsrc4(a.in + (long)z.b * a.H * a.W * a.Cin, (unsigned)a.H * a.W * a.Cin * 4u);   // image b's input, formed where it is used
        const unsigned oobv = 0x80000000u;
        if (lane < 4 * (Y_PW / 2)) {
#pragma unroll
            for (int k = 0; k < Y_PPW; ++k) {
                // piece wave * Y_PPW + k = (patch row, column parity); lane 4 s + quad reads quad `quad` of pixel (row, 2 s + parity). Rows outside the
                // image and the dummy pieces past the 36th go out of range as a whole (uniform select of the lane offset)
                const int pc = wave * Y_PPW + k, row = pc >> 1, gy = z.y0 + row - 1;
                const bool rowok = live && pc < Y_PIECES && gy >= 0 && gy < a.H;
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((pc * XP_PITCH + ((row >> 1) & 3)) * 16));
                const unsigned so_ = __builtin_amdgcn_readfirstlane(rowok ? so + (unsigned)gy * (unsigned)(a.W * a.Cin * 4) : 0u);
#ifdef IM_YABL_NO_PDMA      // -DIM_YABL_* / -DIM_XABL_*: timing-only ablations (wrong results)
                dma16(rs, dst, oobv, so_);
#else
                if (rowok) dma16(rs, dst, (pc & 1) ? z.pv[1] : z.pv[0], so_);     // uniform branch: exactly one of the two is issued, no per-piece lane offset is formed
                else dma16(rs, dst, oobv, so_);
#endif
            }
        }
    };
    // ---- U: an endless stream of quarters (chunk uc, position step uj), the same for every item of this block; wave w moves V row w >> 1,
    // column tile w & 1, its three planes. uw = ring slot written next, ur_ = ring slot read next.
    const unsigned ux_pos = (unsigned)(a.Cout / 32) * 3072u, ux_chunk = 16u * ux_pos;
    const wu32x4 rux = wmake_rsrc4(a.wx, (unsigned)nchunk * ux_chunk);
    const unsigned ux_w = (unsigned)(4 * (wave >> 1)) * ux_pos + (unsigned)(co0 / 32 + (wave & 1)) * 3072u;
    const unsigned ux_lane = (unsigned)lane * 16u;
    int uc = 0, uj = 0, uw = 0, ur_ = 0;
    auto u_dma = [&]() __attribute__((always_inline)) {
        const unsigned so = (unsigned)uc * ux_chunk + (unsigned)uj * ux_pos + ux_w;
        const unsigned dst = ldsR + (unsigned)(uw * Y_QBYTES + wave * 3072);
#pragma unroll
#ifdef IM_YABL_NO_UDMA
        for (int pl = 0; pl < 3; ++pl) dma16(rux, dst + pl * 1024u, 0x80000000u, so);
#else
        for (int pl = 0; pl < 3; ++pl) dma16(rux, dst + pl * 1024u, ux_lane + pl * 1024u, so);
#endif
        uw = uw == Y_NQ - 1 ? 0 : uw + 1;
        if (++uj == 4) { uj = 0; uc = uc + 1 == nchunk ? 0 : uc + 1; }
    };

    // lane (c, hh): tile c of the group's 4 x 8 tile grid, channels 8 hh .. 8 hh + 7 of the chunk
    const int t_ty = c >> 3, t_tx = c & 7;
    const int rowA = ph == 0 ? 0 : (ph == 2 ? 2 : 1), rowB = ph == 0 ? 2 : (ph == 3 ? 3 : (ph == 2 ? 1 : 2));   // wave-uniform
    const int xp_rowA = 8 * g + 2 * t_ty + rowA, xp_rowB = 8 * g + 2 * t_ty + rowB;
    const int xp_A = 2 * xp_rowA * XP_PITCH + ((xp_rowA >> 1) & 3) + t_tx * 4 + 2 * hh, xp_B = 2 * xp_rowB * XP_PITCH + ((xp_rowB >> 1) & 3) + t_tx * 4 + 2 * hh;
    f32x2 rsgn;
    {
        float sg = __uint_as_float(__builtin_amdgcn_readfirstlane(ph == 1 ? 0x3F800000u : 0xBF800000u));
        asm("" : "+s"(sg));
        rsgn = f32x2{sg, sg};
    }
    const f32x2 m1 = minus_one();
    float* const sBias = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + Y_LDS_BYTES);     // the block's 64 biases (it keeps its output channels)
    if (tid < 64) sBias[tid] = a.bias[co0 + tid];
    const float4* const pa = reinterpret_cast<const float4*>(smem);
    const wu32x4* const ring = reinterpret_cast<const wu32x4*>(reinterpret_cast<const char*>(smem) + Y_STAGE_BYTES) + (ph * 2) * 192 + lane;   // + slot * 1536 + (ct * 3 + plane) * 64

    f32x16 acc[8];
    float4 t[2][4];                 // row pass of the next chunk (quarter 2)
    float v[4][8];                  // V row of the chunk in flight
    unsigned ph_[2][4], pm_[2][4], pl_[2][4];
    wu32x4 uf[2][3];

    auto row_reads = [&](int q, int jj, float4& da, float4& db) __attribute__((always_inline)) {
        da = pa[xp_A + (jj & 1) * XP_PITCH + (jj >> 1) * 4 + q];
        db = pa[xp_B + (jj & 1) * XP_PITCH + (jj >> 1) * 4 + q];
    };
    auto col_half = [&](int q) __attribute__((always_inline)) {
        const float4 w0 = sub4(t[q][0], t[q][2], m1), w1 = add4(t[q][1], t[q][2]), w2 = sub4(t[q][2], t[q][1], m1), w3 = sub4(t[q][1], t[q][3], m1);
        v[0][4 * q] = w0.x; v[0][4 * q + 1] = w0.y; v[0][4 * q + 2] = w0.z; v[0][4 * q + 3] = w0.w;
        v[1][4 * q] = w1.x; v[1][4 * q + 1] = w1.y; v[1][4 * q + 2] = w1.z; v[1][4 * q + 3] = w1.w;
        v[2][4 * q] = w2.x; v[2][4 * q + 1] = w2.y; v[2][4 * q + 2] = w2.z; v[2][4 * q + 3] = w2.w;
        v[3][4 * q] = w3.x; v[3][4 * q + 1] = w3.y; v[3][4 * q + 2] = w3.z; v[3][4 * q + 3] = w3.w;
    };
    // the cut of one pair in three pieces (5 + 5 + 1 vector instructions), each pinned where it is written by an empty asm on its values: the
    // optimiser otherwise gathers the cuts where their inputs appear (all of a chunk's cuts landed beside the twelve MFMAs of quarter 3)
#define Y_PIN(x) asm volatile("" : "+v"(x))
    float ra[2], rb[2];
    auto cut_a = [&](int pos, int buf, int i, int k) __attribute__((always_inline)) {
        float x = v[pos][2 * i], y = v[pos][2 * i + 1];
        Y_PIN(x); Y_PIN(y);
#ifdef IM_XABL_NO_CUT
        ph_[buf][i] = __float_as_uint(x); ra[k] = y; rb[k] = x;
#else
        const unsigned h = wcvt_pk(x, y);
        ph_[buf][i] = h;
        ra[k] = x - __uint_as_float(h << 16);
        rb[k] = y - __uint_as_float(h & 0xffff0000u);
#endif
        Y_PIN(ra[k]); Y_PIN(rb[k]);
    };
    auto cut_b = [&](int buf, int i, int k) __attribute__((always_inline)) {
        Y_PIN(ra[k]); Y_PIN(rb[k]);
#ifdef IM_XABL_NO_CUT
        pm_[buf][i] = __float_as_uint(ra[k]);
#else
        const unsigned m = wcvt_pk(ra[k], rb[k]);
        pm_[buf][i] = m;
        ra[k] -= __uint_as_float(m << 16);
        rb[k] -= __uint_as_float(m & 0xffff0000u);
#endif
        Y_PIN(ra[k]); Y_PIN(rb[k]);
    };
    auto cut_l = [&](int buf, int i, int k) __attribute__((always_inline)) {
        Y_PIN(ra[k]);
#ifdef IM_XABL_NO_CUT
        unsigned l = __float_as_uint(rb[k]);
#else
        unsigned l = wcvt_pk(ra[k], rb[k]);
#endif
        Y_PIN(l);
        pl_[buf][i] = l;
    };
    auto uread = [&](int buf, int slot, int ct) __attribute__((always_inline)) {
        const wu32x4* p = ring + slot * 1536 + ct * 192;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) uf[buf][pl] = p[pl * 64];
    };

    // ---- prologue (once per block): the first Y_NQ - 1 quarters of U and the first patch; the transform and the first cut of the first chunk
    YItem cur, nxt;
    setup(item, cur);
    Y_STAMP(0)
#pragma unroll
    for (int q_ = 0; q_ < Y_NQ - 1; ++q_) u_dma();
    patch_dma(cur, 0, true);
    IM_DMA_WAIT();
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { float4 da, db; row_reads(q, jj, da, db); t[q][jj] = sub4(da, db, rsgn); }
    col_half(0); col_half(1);
#pragma unroll
    for (int i = 0; i < 4; ++i) wsplit2(v[0][2 * i], v[0][2 * i + 1], ph_[0][i], pm_[0][i], pl_[0][i]);
    uread(0, 0, 0);
    __syncthreads();                 // every wave has read the stage
    patch_dma(cur, 1, true);
    Y_STAMP(1)

    // ---- quarter (chunk, j): steps (j, ct = 0, 1) with the planes of position j; beside them the cut of position j + 1 (j = 3: of the next chunk's
    // position 0) and, j = 2 / 3, the row / column pass of the next chunk (possibly the next item's first). Transfers complete in issue order;
    // issue order: U(T + 2) at the start of quarter T, the patch two chunks ahead at the start of a quarter 3 behind its U pieces. In front of the
    // barrier that ends quarter T, U(T + 1) (issued at the start of T - 1) must have landed; younger than it:
    //   j = 0: the patch pieces of quarter T - 1 (Y_PPW = 5), U(T + 2) (3)          -> vmcnt(8)   (behind an item boundary the epilogue's stores sit between
    //          the patch pieces and U(T + 2): the wait then also covers all but the last three of them, issued a whole quarter earlier)
    //   j = 1: U(T + 2)                                                             -> vmcnt(3)   (the patch, read in quarter 2, is older than U(T + 1))
    //   j = 2: U(T + 2)                                                             -> vmcnt(3)
    //   j = 3: U(T + 2), this quarter's patch pieces                                -> vmcnt(8)
    int stamp_i = 2;
    auto quarter = [&](auto J_, auto FIRST_, const YItem& pz, int pchunk, bool plive) __attribute__((always_inline)) {
        constexpr int j = decltype(J_)::value;
        constexpr bool FIRST = decltype(FIRST_)::value;
        constexpr int cur_ = j & 1;
        const int slot = ur_;
        ur_ = ur_ == Y_NQ - 1 ? 0 : ur_ + 1;
        u_dma();
        if constexpr (j == 3) patch_dma(pz, pchunk, plive);
        constexpr int nx = (j + 1) & 3;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int s = 2 * j + n;
            const wu32x4 ah = {ph_[cur_][0], ph_[cur_][1], ph_[cur_][2], ph_[cur_][3]}, am = {pm_[cur_][0], pm_[cur_][1], pm_[cur_][2], pm_[cur_][3]},
                         al = {pl_[cur_][0], pl_[cur_][1], pl_[cur_][2], pl_[cur_][3]};
            const wu32x4 bh = uf[n][0], bm = uf[n][1], bl = uf[n][2];
            float4 da[1], db[1];
            f32x16 x = FIRST ? f32x16{} : acc[s];
            // six MFMA slots (h l, l h, m m, h m, m h, h h); behind each its share of the vector work, fenced:
            //   every quarter: the cut of pairs 2 n, 2 n + 1 of the next position (a a b b l l);
            //   j = 2: the row pass of channel half n of the NEXT chunk (the stage holds it since the barrier that ended quarter 1), one pixel
            //          column per slot, its reads one slot ahead; in step 1 also the column pass of half 0;  j = 3, step 0: the column pass of half 1
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                x = mfma_bx(k == 1 ? al : ((k == 2 || k == 4) ? am : ah), k == 0 ? bl : ((k == 2 || k == 3) ? bm : bh), x);
                if (k == 0 && n == 0) uread(1, slot, 1);     // the second step's fragments while the first step's products run
                if (k == 0) cut_a(nx, cur_ ^ 1, 2 * n, 0);
                if (k == 1) cut_a(nx, cur_ ^ 1, 2 * n + 1, 1);
                if (k == 2) cut_b(cur_ ^ 1, 2 * n, 0);
                if (k == 3) cut_b(cur_ ^ 1, 2 * n + 1, 1);
                if (k == 4) { cut_l(cur_ ^ 1, 2 * n, 0); cut_l(cur_ ^ 1, 2 * n + 1, 1); }
                if constexpr (j == 2) {             // pixel column k - 1 behind slot k (read behind slot k - 1): one pair of reads in flight
                    if (k >= 1 && k < 5) t[n][k - 1] = sub4(da[0], db[0], rsgn);
                    if (k < 4) row_reads(n, k, da[0], db[0]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (j == 2) { if (n == 1) col_half(0); }   // (step 1) the column pass of half 0 behind the last slot
            if constexpr (j == 3) { if (n == 0) col_half(1); }
            acc[s] = x;
            __builtin_amdgcn_sched_barrier(0);
        }
        Y_STAMP(stamp_i)
        if constexpr (j == 0 || j == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 + Y_PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        Y_STAMP(stamp_i + 1)
        __syncthreads();
        Y_STAMP(stamp_i + 2)
        stamp_i += 3;
        uread(0, ur_, 0);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    using TT = std::true_type; using FF = std::false_type;
    for (;;) {
        const int nitem = item + (int)gridDim.x;
        const bool has_next = nitem < nitems && region_of(nitem) < ntile;
        if (has_next) setup(nitem, nxt); else nxt = cur;
        // the patch for chunk c + 2 goes out in quarter 3 of chunk c: chunks 2 .. of this item, then chunks 0, 1 of the next
        auto chunk_quarters = [&](int chunk, auto FIRST_) __attribute__((always_inline)) {
            const bool mine = chunk + 2 < nchunk;
            quarter(I0{}, FIRST_, cur, 0, false);
            quarter(I1{}, FIRST_, cur, 0, false);
            quarter(I2{}, FIRST_, cur, 0, false);
            YItem pz;                    // this item's or the next one's patch coordinates, field by field (a reference to one of two structs would
            pz.b = mine ? cur.b : nxt.b; pz.y0 = mine ? cur.y0 : nxt.y0; pz.x0 = mine ? cur.x0 : nxt.x0;      // put both in scratch memory)
            pz.pv[0] = mine ? cur.pv[0] : nxt.pv[0]; pz.pv[1] = mine ? cur.pv[1] : nxt.pv[1];
            quarter(I3{}, FIRST_, pz, mine ? chunk + 2 : chunk + 2 - nchunk, mine || has_next);
        };
        chunk_quarters(0, TT{});          // fresh accumulators
        for (int chunk = 1; chunk < nchunk; ++chunk) chunk_quarters(chunk, FF{});
        Y_STAMP(stamp_i)
#ifdef IM_YABL_NO_EPI
        {
            float keep_ = 0.f;
#pragma unroll
            for (int p_ = 0; p_ < 8; ++p_) keep_ += acc[p_][0] + acc[p_][15];
            if (keep_ == 12345.678f) a.out[0] = keep_;
        }
#else
        {
            // everything the epilogue derives from the lane index is formed HERE, per item: hoisted out of the item loop it would live (in
            // scratch memory) across the whole main loop. The bias comes back from LDS for the same reason.
            int lane_ = lane;
            asm volatile("" : "+v"(lane_));
            const float bias2[2] = {sBias[lane_ & 31], sBias[32 + (lane_ & 31)]};
            wino_epilogue_rounds<POOL>(a, acc, xw, ph, lane_, cur.b, cur.y0 + 8 * g, cur.x0, co0, m1, bias2);
        }
#endif
        Y_STAMP(stamp_i + 1)
        stamp_i += 2;
        if (!has_next) break;
        cur = nxt;
        item = nitem;
    }
    // transfers still in flight (the U stream ran ahead, out-of-range patch pieces) target this block's LDS: drain before the block ends
    IM_DMA_WAIT();
}
#ifdef IM_YSTAMP
extern "C" int im_debug_ystamps(unsigned long long* host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ystamp), n * sizeof(unsigned long long));
}
#endif

template <bool POOL>
static hipError_t launch_bx2(const ConvArgs& a, hipStream_t s) {
    const int ntile = ((a.W + Y_TW - 1) / Y_TW) * ((a.H + Y_TH - 1) / Y_TH) * a.B;
    const int nslices = a.Cout / 64;
    const int nitems = ((ntile + 7) / 8) * 8 * nslices;
    static int n_cu[IM_MAX_DEVICES] = {0};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    if (dev < 0 || dev >= IM_MAX_DEVICES) return hipErrorInvalidDevice;
    if (!n_cu[dev]) {
        if (hipError_t e = hipDeviceGetAttribute(&n_cu[dev], hipDeviceAttributeMultiprocessorCount, dev); e != hipSuccess) return e;
    }
    // one block per CU; a multiple of 8 nslices so that a block keeps its output-channel slice and its residue mod 8 over all its items
    const int unit = 8 * nslices;
    int grid = (n_cu[dev] / unit) * unit;
    if (grid < unit) grid = unit;
    if (grid > nitems) grid = nitems;
    static size_t lds_optin[IM_MAX_DEVICES] = {0};
    if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(&conv3x3_wino_bx2_kernel<POOL>), Y_LDS_BYTES + 256, lds_optin); e != hipSuccess) return e;
    hipLaunchKernelGGL((conv3x3_wino_bx2_kernel<POOL>), dim3(grid), dim3(512), Y_LDS_BYTES + 256, s, a, nitems);
    return hipGetLastError();
}

// plain layers (no fused first layer) with the bf16 planes of U
hipError_t launch_conv3x3_wino_bx2(const ConvArgs& a, hipStream_t s) {
    if (!a.wx || a.img || a.Cin % 16 != 0 || a.Cin < 64 || a.Cout % 64 != 0) return hipErrorInvalidValue;
    return a.pool ? launch_bx2<true>(a, s) : launch_bx2<false>(a, s);
}

}  // namespace im
