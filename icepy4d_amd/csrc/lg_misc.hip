// LightGlue glue kernels: positional encoding, LayerNorm+GELU, token confidence / matchability, early-stop and
// point-pruning decisions (taken on the device: no host round trip per layer), order-preserving compaction,
// and the assignment stage (double log-softmax, mutual nearest neighbour). HBM-bound: wave-per-row reductions,
// coalesced column sweeps; integer atomics only (deterministic).
#include "common.h"
#include "lg_misc.h"

#include "assign_sweep.h"

namespace im {

// ---------------------------------------------------------------------------------------------------------
// normalize_keypoints + LearnableFourierPositionalEncoding (`lightglue/lightglue.py:23-35, 60-74`)
__global__ __launch_bounds__(256) void posenc_kernel(const float* __restrict__ kpts, long kp_bstride, const LGState* __restrict__ st,
                                                      const float* __restrict__ wr, float4 sizes, float* __restrict__ cs,
                                                      float* __restrict__ sn, long enc_bstride) {
    const int b = blockIdx.y;      // image b & 1 of pair b >> 1; every pair of a batch has the same two image sizes
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int i = t >> 5, f = t & 31;
    if (i >= st[b >> 1].n[b & 1]) return;
    const float sw = (b & 1) == 0 ? sizes.x : sizes.z, sh = (b & 1) == 0 ? sizes.y : sizes.w;
    const float scale = fmaxf(sw, sh) / 2.f;
    const float kx = (kpts[(long)b * kp_bstride + i * 2] - sw / 2.f) / scale;
    const float ky = (kpts[(long)b * kp_bstride + i * 2 + 1] - sh / 2.f) / scale;
    const float proj = kx * wr[f * 2] + ky * wr[f * 2 + 1];
    cs[(long)b * enc_bstride + (long)i * 32 + f] = cosf(proj);
    sn[(long)b * enc_bstride + (long)i * 32 + f] = sinf(proj);
}

hipError_t launch_posenc(const float* kpts, long kp_bstride, const LGState* st, int n_images, int n_max, const float* wr,
                         const float* h_size, float* cs, float* sn, long enc_bstride, hipStream_t s) {
    float4 sizes = make_float4(h_size[0], h_size[1], h_size[2], h_size[3]);
    hipLaunchKernelGGL(posenc_kernel, dim3((n_max * 32 + 255) / 256, n_images), dim3(256), 0, s, kpts, kp_bstride, st, wr, sizes, cs, sn, enc_bstride);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm(512, eps 1e-5, affine) + exact-erf GELU in place (`lightglue/lightglue.py:144-149`). Wave per row.
__global__ __launch_bounds__(256) void layernorm_gelu_kernel(float* __restrict__ h, long bstride, const LGState* __restrict__ st,
                                                              const float* __restrict__ g, const float* __restrict__ be) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    if (st[b >> 1].active == 0) return;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= st[b >> 1].n[b & 1]) return;
    float* p = h + (long)b * bstride + (long)row * 512;
    float4 v0 = *reinterpret_cast<float4*>(p + lane * 4);
    float4 v1 = *reinterpret_cast<float4*>(p + 256 + lane * 4);
    const float mean = wave_sum(v0.x + v0.y + v0.z + v0.w + v1.x + v1.y + v1.z + v1.w) * (1.f / 512.f);
    float d[8] = {v0.x - mean, v0.y - mean, v0.z - mean, v0.w - mean, v1.x - mean, v1.y - mean, v1.z - mean, v1.w - mean};
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += d[i] * d[i];
    const float rstd = rsqrtf(wave_sum(ss) * (1.f / 512.f) + 1e-5f);
    const float4 g0 = *reinterpret_cast<const float4*>(g + lane * 4), g1 = *reinterpret_cast<const float4*>(g + 256 + lane * 4);
    const float4 b0 = *reinterpret_cast<const float4*>(be + lane * 4), b1 = *reinterpret_cast<const float4*>(be + 256 + lane * 4);
    const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float y = d[i] * rstd * gg[i] + bb[i];
        o[i] = 0.5f * y * (1.f + erff(y * 0.70710678118654752440f));
    }
    *reinterpret_cast<float4*>(p + lane * 4) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(p + 256 + lane * 4) = make_float4(o[4], o[5], o[6], o[7]);
}

hipError_t launch_layernorm_gelu(float* h, long bstride, const LGState* st, int n_images, int n_max, const float* g, const float* be,
                                 hipStream_t s) {
    hipLaunchKernelGGL(layernorm_gelu_kernel, dim3((n_max + 3) / 4, n_images), dim3(256), 0, s, h, bstride, st, g, be);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Row dot products with up to two 256-vectors (wave per row):
//   out0 = act0(x . w0 + b0)   token confidence (`lightglue/lightglue.py:77-89`) or matchability logit (`:281-282`)
//   out1 = sigmoid(x . w1 + b1) matchability (`:284-285`)
// and an integer count of rows with out0 < thr (the early-stop statistic, `:571-579`).
static constexpr int RD_ROWS = 8;  // rows per wave: one counter atomic per block of 32 rows instead of one per row

__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ x, long bstride, LGState* __restrict__ st,
                                                      const float* __restrict__ w0, const float* __restrict__ b0, int act0,
                                                      const float* __restrict__ w1, const float* __restrict__ b1,
                                                      const int* __restrict__ sel, float* __restrict__ out0,
                                                      float* __restrict__ out1, long out_bstride, float thr,
                                                      int count_layer, int check_active) {
    __shared__ int blk_cnt;
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    LGState* sp = st + (b >> 1);
    if (check_active && sp->active == 0) return;
    int* counter = count_layer >= 0 ? &sp->cnt[count_layer] : nullptr;
    const int n = sp->n[b & 1];
    const int row0 = (blockIdx.x * 4 + wave) * RD_ROWS;
    if (blockIdx.x * 4 * RD_ROWS >= n) return;  // block-uniform
    if (threadIdx.x == 0) blk_cnt = 0;
    __syncthreads();
    const int l = sel ? sel[b >> 1] : 0;
    float4 wa = make_float4(0.f, 0.f, 0.f, 0.f), wb = wa;
    float ba = 0.f, bb = 0.f;
    if (w0) { wa = *reinterpret_cast<const float4*>(w0 + (long)l * 256 + lane * 4); ba = b0[l]; }
    if (w1) { wb = *reinterpret_cast<const float4*>(w1 + (long)l * 256 + lane * 4); bb = b1[l]; }
    int cnt = 0;
    float4 vr[RD_ROWS];      // all rows of the wave in flight before the first reduction
#pragma unroll
    for (int i = 0; i < RD_ROWS; ++i)
        vr[i] = row0 + i < n ? *reinterpret_cast<const float4*>(x + (long)b * bstride + (long)(row0 + i) * 256 + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < RD_ROWS; ++i) {
        const int row = row0 + i;
        if (row >= n) break;  // wave-uniform
        const float4 v = vr[i];
        if (w0) {
            float d = wave_sum(v.x * wa.x + v.y * wa.y + v.z * wa.z + v.w * wa.w) + ba;
            if (act0) d = sigmoidf(d);
            if (lane == 0) {
                out0[(long)b * out_bstride + row] = d;
                if (d < thr) ++cnt;
            }
        }
        if (w1) {
            const float d = sigmoidf(wave_sum(v.x * wb.x + v.y * wb.y + v.z * wb.z + v.w * wb.w) + bb);
            if (lane == 0) out1[(long)b * out_bstride + row] = d;
        }
    }
    if (counter) {
        if (lane == 0 && cnt) atomicAdd(&blk_cnt, cnt);
        __syncthreads();
        if (threadIdx.x == 0 && blk_cnt) atomicAdd(counter, blk_cnt);
    }
}

hipError_t launch_rowdot(const float* x, long bstride, LGState* st, int n_images, int n_max, const float* w0, const float* b0, int act0,
                         const float* w1, const float* b1, const int* sel, float* out0, float* out1, long out_bstride,
                         float thr, int count_layer, int check_active, hipStream_t s) {
    hipLaunchKernelGGL(rowdot_kernel, dim3((n_max + 4 * RD_ROWS - 1) / (4 * RD_ROWS), n_images), dim3(256), 0, s, x, bstride, st, w0, b0, act0,
                       w1, b1, sel, out0, out1, out_bstride, thr, count_layer, check_active);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Early stop (`check_if_stop`, `lightglue/lightglue.py:571-579`) and point pruning (`get_pruning_mask` +
// `torch.where` compaction, `:495-510, 563-569`) decided on the device. One block handles both images in turn.
// When the matcher is (or becomes) inactive it emits identity keep lists so that the static ping-pong of the
// descriptor buffers stays valid.
__global__ __launch_bounds__(1024) void stop_prune_kernel(LGState* __restrict__ st, int layer, int do_stop, int do_prune,
                                                           float depth_conf, float keep_thr, float conf_thr,
                                                           const float* __restrict__ conf, const float* __restrict__ msc,
                                                           long vec_bstride, const int* __restrict__ ind_cur,
                                                           int* __restrict__ ind_next, int* __restrict__ keep_idx,
                                                           int* __restrict__ prune, long idx_bstride, int prune_min) {
    __shared__ int part[1024];
    __shared__ int sh_active;
    const int tid = threadIdx.x;
    const int pair = blockIdx.x, b = blockIdx.y;     // one block per image; both blocks of a pair take the same stop decision
    st += pair;
    conf += 2 * pair * vec_bstride; msc += 2 * pair * vec_bstride;
    ind_cur += 2 * pair * idx_bstride; ind_next += 2 * pair * idx_bstride; keep_idx += 2 * pair * idx_bstride; prune += 2 * pair * idx_bstride;
    if (tid == 0) {
        int act = st->active;
        if (act && do_stop) {
            const float ratio = 1.0f - (float)st->cnt[layer] / (float)(st->n_orig[0] + st->n_orig[1]);
            if (ratio > depth_conf) act = 0;
        }
        sh_active = act;
    }
    __syncthreads();
    {
        const int n = st->n[b];
        // `if do_point_pruning and desc.shape[-2] > pruning_th` (`lightglue.py:495, 503`): per image, on its live count
        const bool live = sh_active != 0 && do_prune && n > prune_min;
        const int per = (n + 1023) / 1024;
        const int lo = min(tid * per, n), hi = min(lo + per, n);
        const float* cf = conf + (long)b * vec_bstride;
        const float* ms = msc + (long)b * vec_bstride;
        int cnt = 0;
        for (int e = lo; e < hi; ++e) {
            bool keep = true;
            if (live) {
                keep = ms[e] > keep_thr;
                if (do_stop) keep = keep || (cf[e] <= conf_thr);
            }
            cnt += keep;
        }
        part[tid] = cnt;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            int v = (tid >= off) ? part[tid - off] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        int pos = part[tid] - cnt;
        const int total = part[1023];
        const int* ic = ind_cur + (long)b * idx_bstride;
        int* in = ind_next + (long)b * idx_bstride;
        int* ki = keep_idx + (long)b * idx_bstride;
        int* pr = prune + (long)b * idx_bstride;
        for (int e = lo; e < hi; ++e) {
            bool keep = true;
            if (live) {
                keep = ms[e] > keep_thr;
                if (do_stop) keep = keep || (cf[e] <= conf_thr);
            }
            if (keep) {
                const int orig = ic[e];
                ki[pos] = e;
                in[pos] = orig;
                if (live) pr[orig] += 1;
                ++pos;
            }
        }
        __syncthreads();
        if (tid == 0) st->n[b] = total;
    }
}

// the stop decision itself, published after both image blocks of every pair have read the old state (next launch on the stream)
__global__ void stop_commit_kernel(LGState* __restrict__ st, int n_pairs, int layer, float depth_conf) {
    const int p = threadIdx.x;
    if (p >= n_pairs) return;
    st += p;
    if (st->active) {
        const float ratio = 1.0f - (float)st->cnt[layer] / (float)(st->n_orig[0] + st->n_orig[1]);
        if (ratio > depth_conf) { st->active = 0; st->stop_layer = layer; }
    }
}

// do_stop without do_prune: the stop decision is committed by its own tiny launch; with pruning the gather kernel that follows
// commits it (launch_gather_rows(commit_layer = layer))
hipError_t launch_stop_prune(LGState* st, int n_pairs, int layer, int do_stop, int do_prune, float depth_conf, float keep_thr,
                             float conf_thr, const float* conf, const float* msc, long vec_bstride, const int* ind_cur,
                             int* ind_next, int* keep_idx, int* prune, long idx_bstride, int prune_min, hipStream_t s) {
    hipLaunchKernelGGL(stop_prune_kernel, dim3(n_pairs, 2), dim3(1024), 0, s, st, layer, do_stop, do_prune, depth_conf, keep_thr, conf_thr,
                       conf, msc, vec_bstride, ind_cur, ind_next, keep_idx, prune, idx_bstride, prune_min);
    if (do_stop && !do_prune) hipLaunchKernelGGL(stop_commit_kernel, dim3(1), dim3(64), 0, s, st, n_pairs, layer, depth_conf);
    return hipGetLastError();
}

// dst[p] = src[keep_idx[p]] for descriptor rows (256) and rotary tables (2 x 32); wave per row
__global__ __launch_bounds__(256) void gather_rows_kernel(const LGState* __restrict__ st, const int* __restrict__ keep_idx,
                                                           long idx_bstride, const float* __restrict__ x_src,
                                                           float* __restrict__ x_dst, long x_bstride,
                                                           const float* __restrict__ cs_src, float* __restrict__ cs_dst,
                                                           const float* __restrict__ sn_src, float* __restrict__ sn_dst,
                                                           long enc_bstride, LGState* __restrict__ st_commit, int n_pairs, int layer,
                                                           float depth_conf) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    // the early-stop decision of this layer is published here (every stop_prune block has read the old flag by now: it ran
    // in the previous launch of the stream; nothing in this kernel reads it)
    if (st_commit && blockIdx.x == 0 && b == 0 && (int)threadIdx.x < n_pairs) {
        LGState* sp = st_commit + threadIdx.x;
        if (sp->active) {
            const float ratio = 1.0f - (float)sp->cnt[layer] / (float)(sp->n_orig[0] + sp->n_orig[1]);
            if (ratio > depth_conf) { sp->active = 0; sp->stop_layer = layer; }
        }
    }
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= st[b >> 1].n[b & 1]) return;
    const int e = keep_idx[(long)b * idx_bstride + p];
    *reinterpret_cast<float4*>(x_dst + (long)b * x_bstride + (long)p * 256 + lane * 4) =
        *reinterpret_cast<const float4*>(x_src + (long)b * x_bstride + (long)e * 256 + lane * 4);
    if (lane < 8)
        *reinterpret_cast<float4*>(cs_dst + (long)b * enc_bstride + (long)p * 32 + lane * 4) =
            *reinterpret_cast<const float4*>(cs_src + (long)b * enc_bstride + (long)e * 32 + lane * 4);
    else if (lane < 16)
        *reinterpret_cast<float4*>(sn_dst + (long)b * enc_bstride + (long)p * 32 + (lane - 8) * 4) =
            *reinterpret_cast<const float4*>(sn_src + (long)b * enc_bstride + (long)e * 32 + (lane - 8) * 4);
}

hipError_t launch_gather_rows(LGState* st, int n_images, int n_max, const int* keep_idx, long idx_bstride, const float* x_src,
                              float* x_dst, long x_bstride, const float* cs_src, float* cs_dst, const float* sn_src,
                              float* sn_dst, long enc_bstride, int commit_layer, float depth_conf, hipStream_t s) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3((n_max + 3) / 4, n_images), dim3(256), 0, s, st, keep_idx, idx_bstride, x_src, x_dst,
                       x_bstride, cs_src, cs_dst, sn_src, sn_dst, enc_bstride, commit_layer >= 0 ? st : nullptr, n_images / 2, commit_layer,
                       depth_conf);
    return hipGetLastError();
}

__global__ void lg_init_kernel(LGState* st, const int* __restrict__ n_in, int* __restrict__ ind, int* __restrict__ prune,
                               long idx_bstride, int n_max, int* __restrict__ out_m, float* __restrict__ out_s, long out_bstride) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        LGState* sp = st + (b >> 1);
        const int n = min(n_in[b], n_max);
        sp->n[b & 1] = n;
        sp->n_orig[b & 1] = n;
        if ((b & 1) == 0) {
            sp->active = 1;
            sp->stop_layer = -1;
            for (int l = 0; l < 16; ++l) sp->cnt[l] = 0;
        }
    }
    if (i < n_max) {
        ind[(long)b * idx_bstride + i] = i;
        prune[(long)b * idx_bstride + i] = 1;
        out_m[(long)b * out_bstride + i] = -1;
        out_s[(long)b * out_bstride + i] = 0.f;
    }
}

hipError_t launch_lg_init(LGState* st, int n_images, const int* n_in, int* ind, int* prune, long idx_bstride, int n_max, int* out_m,
                          float* out_s, long out_bstride, hipStream_t s) {
    hipLaunchKernelGGL(lg_init_kernel, dim3((n_max + 255) / 256, n_images), dim3(256), 0, s, st, n_in, ind, prune, idx_bstride, n_max, out_m, out_s, out_bstride);
    return hipGetLastError();
}

// sel = last executed layer: stop_layer if the matcher stopped early, else n_layers - 1 (`lightglue.py:512-513`)
__global__ void lg_select_layer_kernel(LGState* st, int n_pairs, int n_layers, int* sel, int* info) {
    const int p = threadIdx.x;
    if (p >= n_pairs) return;
    st += p;
    const int last = st->stop_layer >= 0 ? st->stop_layer : n_layers - 1;
    sel[p] = last;
    info[4 * p] = last + 1;
    info[4 * p + 1] = st->n[0];
    info[4 * p + 2] = st->n[1];
    info[4 * p + 3] = 0;
    st->active = 1;  // the assignment stage always runs
}

hipError_t launch_lg_select_layer(LGState* st, int n_pairs, int n_layers, int* sel, int* info, hipStream_t s) {
    hipLaunchKernelGGL(lg_select_layer_kernel, dim3(1), dim3(64), 0, s, st, n_pairs, n_layers, sel, info);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Assignment (`sigmoid_log_double_softmax` + `filter_matches`, `lightglue/lightglue.py:253-306`) on a
// materialised similarity matrix sim [m][ld].
//   score(i, j) = ((x - rmax_i) - rlog_i) + ((x - cmax_j) - clog_j) + (lz0_i + lz1_j)
// (same association as the reference: log_softmax rows + log_softmax columns, then + certainties).

// Two sweeps over sim, each reading it ONCE for both directions (16 rows per block, a wave owns 1024-column chunks, 16-byte
// loads; rows reduce across the lanes of a wave, columns inside a lane over the block's rows):
//   lse_stats  : row (max, log sum exp) complete per block; column (max, sum) partials per 16-row strip -> col_lse_combine
//   best_sweep : with both normalisers known, row arg-max complete per block; column arg-max partials per strip -> col_best_combine
// (AS_* constants, for_pair, load_row16, fexp, assign_score: assign_sweep.h, shared with the one-pair-per-launch forms in lg_assign_pipe.hip)

// The running (max, sum) of the strip's 16 rows live in LDS between row groups ([row][thread]: every thread touches its own slots only,
// no synchronisation), so that the row groups are a real loop - four rows of loads in flight, then their arithmetic - instead of 16 rows
// of registers: 320 -> 118 registers (four waves per SIMD instead of one), same operations in the same order (results bit-identical).
template <bool VEC>
__global__ __launch_bounds__(256, 3) void lse_stats_kernel(AssignArgs aa) {      // (four blocks per CU = 128 registers: 5 of them spilled since the
                                                                                  // row group's 16 loads are issued together, round 5)
    __shared__ float2 rs[AS_ROWS][4];
    __shared__ float2 rst[AS_ROWS][256];
    const AssignArgs a = for_pair(aa, blockIdx.y);
    const float* __restrict__ sim = a.sim;
    const int ld = a.ld, kmax = a.n_max;
    float* __restrict__ rmax = a.rmax; float* __restrict__ rlog = a.rlog;
    float2* __restrict__ cpart = a.part;
    const int m = *a.m_ptr, n = *a.n_ptr;
    const int i0 = blockIdx.x * AS_ROWS;
    if (i0 >= m || n <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nrow = min(AS_ROWS, m - i0);
#pragma unroll
    for (int r = 0; r < AS_ROWS; ++r) rst[r][tid] = make_float2(AS_NEG, 0.f);
    for (int c0 = wave * AS_CHUNK; c0 < n; c0 += 4 * AS_CHUNK) {
        float cM[16], cS[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) { cM[e] = AS_NEG; cS[e] = 0.f; }
#pragma unroll 1
        for (int g = 0; g < AS_ROWS / 4; ++g) {
            float x[4][16];
            // rows past the strip's end: clamped address, every column masked
            load_rows16<VEC, 4>([&](int rr) { return sim + (long)min(i0 + 4 * g + rr, m - 1) * ld; },
                                [&](int rr) { return 4 * g + rr < nrow ? n : 0; }, c0, lane, x);
            // rows: one online step per row with the lane's 16 entries
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const float2 st = rst[4 * g + rr][tid];
                float mx = x[rr][0];
#pragma unroll
                for (int e = 1; e < 16; ++e) mx = fmaxf(mx, x[rr][e]);
                const float nm = fmaxf(st.x, mx);
                float acc = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc += fexp(x[rr][e] - nm);
                rst[4 * g + rr][tid] = make_float2(nm, st.y * fexp(st.x - nm) + acc);
            }
            // columns: one online step per column with these four rows
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float mx = fmaxf(fmaxf(x[0][e], x[1][e]), fmaxf(x[2][e], x[3][e]));
                const float nm = fmaxf(cM[e], mx);
                cS[e] = cS[e] * fexp(cM[e] - nm) + ((fexp(x[0][e] - nm) + fexp(x[1][e] - nm)) + (fexp(x[2][e] - nm) + fexp(x[3][e] - nm)));
                cM[e] = nm;
            }
        }
        // the strip's column partials: a lane's four columns of a quad are 32 contiguous bytes, the wave's 2 KB - two 16-byte stores per
        // quad (eight 8-byte stores scattered over the same 2 KB before: a quarter of every store instruction's bytes)
        float2* cp = cpart + (long)blockIdx.x * kmax;
        const bool pvec = (kmax & 1) == 0;          // 16-byte aligned strip rows (block-uniform)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = c0 + q * 256 + lane * 4;
            if (pvec && j + 3 < n) {
                float4* d = reinterpret_cast<float4*>(cp + j);
                d[0] = make_float4(cM[4 * q], cS[4 * q], cM[4 * q + 1], cS[4 * q + 1]);
                d[1] = make_float4(cM[4 * q + 2], cS[4 * q + 2], cM[4 * q + 3], cS[4 * q + 3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (j + e < n) cp[j + e] = make_float2(cM[4 * q + e], cS[4 * q + e]);   // rows >= nrow contributed exp(AS_NEG - max) = 0
            }
        }
    }
#pragma unroll 4
    for (int r = 0; r < AS_ROWS; ++r) {
        const float2 st = rst[r][tid];
        const float M = wave_max(st.x);
        const float S = wave_sum(st.y * fexp(st.x - M));
        if (lane == 0) rs[r][wave] = make_float2(M, S);
    }
    __syncthreads();
    if (tid < nrow) {
        const float2 a = rs[tid][0], b = rs[tid][1], c = rs[tid][2], d = rs[tid][3];
        const float M = fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x));
        const float S = (a.y * fexp(a.x - M) + b.y * fexp(b.x - M)) + (c.y * fexp(c.x - M) + d.y * fexp(d.x - M));
        rmax[i0 + tid] = M;
        rlog[i0 + tid] = logf(S);
    }
}

// column normalisers from the strip partials: 32 columns x 8 strip groups per block, loads issued eight at a time
static constexpr int CC_COLS = 32, CC_GROUPS = 8;

__global__ __launch_bounds__(256) void col_lse_combine_kernel(AssignArgs aa) {
    // Products are rounded, then added - never fused. Whether `a + b * c` becomes one fused operation is otherwise the compiler's choice per
    // code shape: with the loads behind branches (rounds 3-4) nothing here was fused, with the grouped loads below five of the eight products
    // were - a last-bit change of clog and, through it, of every matching score. (`__fmul_rn` / `__fadd_rn` do not prevent it on this toolchain.)
#pragma clang fp contract(off)
    __shared__ float2 red[CC_GROUPS][CC_COLS];
    const AssignArgs a = for_pair(aa, blockIdx.y);
    const float2* __restrict__ cpart = a.part;
    const int kmax = a.n_max;
    float* __restrict__ cmax = a.cmax; float* __restrict__ clog = a.clog;
    const int m = *a.m_ptr, n = *a.n_ptr;
    const int c = threadIdx.x & (CC_COLS - 1), g = threadIdx.x / CC_COLS;
    const int j = blockIdx.x * CC_COLS + c;
    if (blockIdx.x * CC_COLS >= n) return;
    const int ns = (m + AS_ROWS - 1) / AS_ROWS;
    float M = AS_NEG, S = 0.f;
    if (j < n)
        for (int s0 = g; s0 < ns; s0 += 8 * CC_GROUPS) {
            float2 p[8];      // eight loads in flight: clamped address + fence + mask (behind `sidx < ns ? .. : ..` each load was a branch with its own wait)
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] = cpart[(long)min(s0 + u * CC_GROUPS, ns - 1) * kmax + j];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u * CC_GROUPS >= ns) p[u] = make_float2(AS_NEG, 0.f);
            float mx = p[0].x;
#pragma unroll
            for (int u = 1; u < 8; ++u) mx = fmaxf(mx, p[u].x);
            const float nm = fmaxf(M, mx);
            float acc = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += p[u].y * fexp(p[u].x - nm);
            S = S * fexp(M - nm) + acc;
            M = nm;
        }
    red[g][c] = make_float2(M, S);
    __syncthreads();
    if (g == 0 && j < n) {
        float Mx = red[0][c].x;
#pragma unroll
        for (int u = 1; u < CC_GROUPS; ++u) Mx = fmaxf(Mx, red[u][c].x);
        float Sx = 0.f;
#pragma unroll
        for (int u = 0; u < CC_GROUPS; ++u) Sx += red[u][c].y * fexp(red[u][c].x - Mx);
        cmax[j] = Mx;
        clog[j] = logf(Sx);
    }
}

__global__ __launch_bounds__(256) void logsig_kernel(const float* __restrict__ z, long bstride, const LGState* __restrict__ st,
                                                      float* __restrict__ lz) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i < st[b >> 1].n[b & 1]) lz[(long)b * bstride + i] = log_sigmoid(z[(long)b * bstride + i]);
}

// One sweep: row arg-max (first column among ties: torch.max semantics) complete per block; per-strip column arg-max keys
// (ordered score bits, ~row): a larger key = a larger score or, at equal score, a lower row.
template <int MODE, bool VEC>
__global__ __launch_bounds__(256, 3) void best_sweep_kernel(AssignArgs aa) {
    __shared__ float rb_v[AS_ROWS][4];
    __shared__ int rb_j[AS_ROWS][4];
    __shared__ float sh_rm[AS_ROWS], sh_rl[AS_ROWS], sh_l0[AS_ROWS];
    __shared__ float pbv[AS_ROWS][256];     // the lanes' running row maxima between row groups (as lse_stats_kernel's rst)
    __shared__ int pbj[AS_ROWS][256];
    const AssignArgs a = for_pair(aa, blockIdx.y);
    const float* __restrict__ sim = a.sim;
    const int ld = a.ld, kmax = a.n_max;
    const float* __restrict__ rmax = a.rmax; const float* __restrict__ rlog = a.rlog;
    const float* __restrict__ cmax = a.cmax; const float* __restrict__ clog = a.clog;
    const float* __restrict__ lz0 = a.lz0; const float* __restrict__ lz1 = a.lz1;
    int* __restrict__ ridx = a.ridx; float* __restrict__ rval = a.rval;
    unsigned long long* __restrict__ cbpart = reinterpret_cast<unsigned long long*>(a.part);
    const int m = *a.m_ptr, n = *a.n_ptr;
    const int i0 = blockIdx.x * AS_ROWS;
    if (i0 >= m || n <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nrow = min(AS_ROWS, m - i0);
    if (tid < AS_ROWS) {
        const int i = min(i0 + tid, m - 1);
        sh_rm[tid] = rmax[i];
        sh_rl[tid] = MODE == 0 ? rlog[i] : rlog[0];
        sh_l0[tid] = MODE == 0 ? lz0[i] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < AS_ROWS; ++r) { pbv[r][tid] = -INFINITY; pbj[r][tid] = 0x7fffffff; }
    __syncthreads();
    for (int c0 = wave * AS_CHUNK; c0 < n; c0 += 4 * AS_CHUNK) {
        float cm[16], cl[16], l1[16], cbv[16];
        int cbi[16];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = min(c0 + q * 256 + lane * 4 + e, n - 1);
                cm[4 * q + e] = cmax[j];
                cl[4 * q + e] = MODE == 0 ? clog[j] : 0.f;
                l1[4 * q + e] = MODE == 0 ? lz1[j] : 0.f;
                cbv[4 * q + e] = -INFINITY;
                cbi[4 * q + e] = -1;
            }
#pragma unroll 1
        for (int g = 0; g < AS_ROWS / 2; ++g) {      // two rows per trip: 155 registers = three waves per SIMD (four rows: 187 = two)
            if (2 * g >= nrow) break;     // block-uniform
            float x[2][16];      // two rows = 8 sixteen-byte loads in flight per lane
            load_rows16<VEC, 2>([&](int rr) { return sim + (long)min(i0 + 2 * g + rr, m - 1) * ld; }, [&](int) { return n; }, c0, lane, x);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int r = 2 * g + rr;
                if (r >= nrow) break;     // block-uniform
                const float rm = sh_rm[r], rl = sh_rl[r], l0 = sh_l0[r];
                float bv = pbv[r][tid];
                int bj = pbj[r][tid];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int j = c0 + q * 256 + lane * 4 + e;
                        const int k = 4 * q + e;
                        const float v = assign_score<MODE>(x[rr][k], rm, rl, cm[k], cl[k], l0, l1[k]);
                        if (j < n) {      // (as selects instead of branches this kernel spills 1,700 registers at its 168-register budget; the
                            // one-pair form in lg_assign_pipe.hip, which has the whole register file, is branch-free)
                            if (v > bv || bj == 0x7fffffff) { bv = v; bj = j; }      // ascending j in a lane's visit order
                            if (v > cbv[k] || cbi[k] < 0) { cbv[k] = v; cbi[k] = i0 + r; }
                        }
                    }
                pbv[r][tid] = bv;
                pbj[r][tid] = bj;
            }
        }
        unsigned long long* cb = cbpart + (long)blockIdx.x * kmax;
        const bool pvec = (kmax & 1) == 0;          // as in lse_stats_kernel: two 16-byte stores per quad
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = c0 + q * 256 + lane * 4;
            unsigned long long key[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                key[e] = ((unsigned long long)f2ord(cbv[4 * q + e]) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)cbi[4 * q + e]);
            if (pvec && j + 3 < n) {
                ulonglong2* d = reinterpret_cast<ulonglong2*>(cb + j);
                d[0] = make_ulonglong2(key[0], key[1]);
                d[1] = make_ulonglong2(key[2], key[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (j + e < n) cb[j + e] = key[e];
            }
        }
    }
    // rows: lanes / chunks visit columns out of order, so ties resolve on the column index explicitly
#pragma unroll 4
    for (int r = 0; r < AS_ROWS; ++r) {
        float v = pbv[r][tid];
        int j = pbj[r][tid];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(v, off);
            const int oj = __shfl_xor(j, off);
            if (ov > v || (ov == v && oj < j)) { v = ov; j = oj; }
        }
        if (lane == 0) { rb_v[r][wave] = v; rb_j[r][wave] = j; }
    }
    __syncthreads();
    if (tid < nrow) {
        float v = rb_v[tid][0];
        int j = rb_j[tid][0];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float ov = rb_v[tid][w];
            const int oj = rb_j[tid][w];
            if (ov > v || (ov == v && oj < j)) { v = ov; j = oj; }
        }
        ridx[i0 + tid] = j;
        rval[i0 + tid] = v;
    }
}

__global__ __launch_bounds__(256) void col_best_combine_kernel(AssignArgs aa) {
    __shared__ unsigned long long red[CC_GROUPS][CC_COLS];
    const AssignArgs a = for_pair(aa, blockIdx.y);
    const unsigned long long* __restrict__ cbpart = reinterpret_cast<const unsigned long long*>(a.part);
    const int kmax = a.n_max;
    unsigned long long* __restrict__ cbest = a.cbest;
    const int m = *a.m_ptr, n = *a.n_ptr;
    const int c = threadIdx.x & (CC_COLS - 1), g = threadIdx.x / CC_COLS;
    const int j = blockIdx.x * CC_COLS + c;
    if (blockIdx.x * CC_COLS >= n) return;
    const int ns = (m + AS_ROWS - 1) / AS_ROWS;
    unsigned long long best = 0ull;
    if (j < n)
        for (int s0 = g; s0 < ns; s0 += 8 * CC_GROUPS) {
            unsigned long long k[8];      // as in col_lse_combine_kernel: eight loads in flight
#pragma unroll
            for (int u = 0; u < 8; ++u) k[u] = cbpart[(long)min(s0 + u * CC_GROUPS, ns - 1) * kmax + j];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u * CC_GROUPS >= ns) k[u] = 0ull;
#pragma unroll
            for (int u = 0; u < 8; ++u) best = k[u] > best ? k[u] : best;
        }
    red[g][c] = best;
    __syncthreads();
    if (g == 0 && j < n) {
        unsigned long long bb = red[0][c];
#pragma unroll
        for (int u = 1; u < CC_GROUPS; ++u) bb = red[u][c] > bb ? red[u][c] : bb;
        cbest[j] = bb;
    }
}

// mutual check + threshold (`filter_matches`), then scatter to the original index space (`lightglue.py:528-539`)
__global__ __launch_bounds__(256) void filter_scatter_kernel(AssignArgs aa) {
    const AssignArgs a = for_pair(aa, blockIdx.z);
    const int* __restrict__ ridx = a.ridx; const float* __restrict__ rval = a.rval;
    const unsigned long long* __restrict__ cbest = a.cbest;
    const float th = a.threshold;
    const int* __restrict__ ind0 = a.ind0; const int* __restrict__ ind1 = a.ind1;
    int* __restrict__ out_m0 = a.out_m0; int* __restrict__ out_m1 = a.out_m1;
    float* __restrict__ out_s0 = a.out_s0; float* __restrict__ out_s1 = a.out_s1;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int m = *a.m_ptr, n = *a.n_ptr;
    if (m <= 0 || n <= 0) return;  // outputs keep their -1 / 0 initialisation (`superglue.py:255-262`)
    auto col_of = [&](int j) { return (int)(0xFFFFFFFFu - (unsigned)(cbest[j] & 0xFFFFFFFFull)); };
    if (blockIdx.y == 0) {
        if (t >= m) return;
        const int j = ridx[t];
        const bool mutual = col_of(j) == t;
        const float ms = mutual ? expf(rval[t]) : 0.f;
        const bool valid = mutual && ms > th;
        const int o = ind0 ? ind0[t] : t;
        out_m0[o] = valid ? (ind1 ? ind1[j] : j) : -1;
        out_s0[o] = ms;
    } else {
        if (t >= n) return;
        const int i = col_of(t);
        const bool mutual1 = ridx[i] == t;
        const bool mutual0 = mutual1;  // col_of(ridx[i]) == i  <=>  ridx[i] == t when i = col_of(t)
        const float ms0 = mutual0 ? expf(rval[i]) : 0.f;
        const float ms1 = mutual1 ? ms0 : 0.f;
        const bool valid1 = mutual1 && (mutual0 && ms0 > th);
        const int o = ind1 ? ind1[t] : t;
        out_m1[o] = valid1 ? (ind0 ? ind0[i] : i) : -1;
        out_s1[o] = ms1;
    }
}

// Zero-fill by a kernel, not hipMemsetAsync: a memset NODE inside a captured HIP graph (ROCm 7.2) faulted when that graph
// was replayed after other graphs had been captured on the same buffers (reproduced with three whole-pair graphs of
// different image shapes: the 4th call, a replay of the first graph, took a GPU memory fault); kernel nodes do not have the problem.
__global__ void zero_words_kernel(unsigned* __restrict__ p, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 0u;
}

hipError_t launch_zero_words(void* p, long nwords, hipStream_t s) {
    if (nwords <= 0) return hipSuccess;
    const long nb = (nwords + 255) / 256;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)(nb > 1024 ? 1024 : nb)), dim3(256), 0, s, reinterpret_cast<unsigned*>(p), nwords);
    return hipGetLastError();
}

hipError_t launch_assign(const AssignArgs& a, hipStream_t s) {
    const int kr = a.m_max, kc = a.n_max, P = a.n_pairs > 0 ? a.n_pairs : 1;
    const int nstrips = (kr + AS_ROWS - 1) / AS_ROWS;
    const bool vec = (a.ld % 4) == 0 && (reinterpret_cast<uintptr_t>(a.sim) % 16) == 0 && (a.sim_ps % 4) == 0;
    const dim3 gs(nstrips, P), gc((kc + CC_COLS - 1) / CC_COLS, P);
    // one pair per launch and at most ~one strip block per CU (4096 keypoints: 256 strips): the software-pipelined forms of the two sweeps
    // (bit-identical; they take the whole register file, so only where there is one wave per SIMD anyway)
    const bool pipe = vec && P == 1 && nstrips <= 384;
    if (a.mode == 0) {
        if (pipe) launch_lse_stats_pipe(a, gs, s);
        else if (vec) hipLaunchKernelGGL(lse_stats_kernel<true>, gs, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(lse_stats_kernel<false>, gs, dim3(256), 0, s, a);
        hipLaunchKernelGGL(col_lse_combine_kernel, gc, dim3(256), 0, s, a);
        if (pipe) launch_best_sweep_pipe(a, gs, s);
        else if (vec) hipLaunchKernelGGL((best_sweep_kernel<0, true>), gs, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((best_sweep_kernel<0, false>), gs, dim3(256), 0, s, a);
    } else {  // optimal transport: rmax = u, cmax = v, rlog[0] = norm were produced by the Sinkhorn sweeps
        if (pipe) launch_best_sweep_pipe(a, gs, s);
        else if (vec) hipLaunchKernelGGL((best_sweep_kernel<1, true>), gs, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((best_sweep_kernel<1, false>), gs, dim3(256), 0, s, a);
    }
    hipLaunchKernelGGL(col_best_combine_kernel, gc, dim3(256), 0, s, a);
    const int kk = kr > kc ? kr : kc;
    hipLaunchKernelGGL(filter_scatter_kernel, dim3((kk + 255) / 256, 2, P), dim3(256), 0, s, a);
    return hipGetLastError();
}

// match-table record of one pair (layout in icepy4d_amd/sequence.py): header {epoch, n0, n1, n_matches, stop, 0, 0, 0},
// matches0 [K], matching_scores0 [K] (bit patterns), and - when kpts is given - the keypoints of both images [2][K][2] as bit
// patterns (the 98 KB record of SURVEY 8d config 4). One block per pair; the match count is an integer block reduction.
__global__ __launch_bounds__(256) void pack_record_kernel(const int* __restrict__ n, const int* __restrict__ matches0,
                                                           const float* __restrict__ mscores0, const int* __restrict__ info,
                                                           int epoch, int K, int* __restrict__ rec, const float* __restrict__ kpts) {
    __shared__ int red[4];
    {   // pair blockIdx.x of a batch: n [2P], matches / scores [2P][K] (row 2p = matches0 of pair p), info [P][4], kpts [2P][K][2]
        const int p = blockIdx.x;
        n += 2 * p; matches0 += (long)2 * p * K; mscores0 += (long)2 * p * K; info += 4 * p; epoch += p;
        rec += (long)p * (8 + (kpts ? 6 : 2) * K);
        if (kpts) kpts += (long)4 * p * K;
    }
    int cnt = 0;
    for (int i = threadIdx.x; i < K; i += 256) {
        const int m = matches0[i];
        rec[8 + i] = m;
        rec[8 + K + i] = __float_as_int(mscores0[i]);
        cnt += m > -1;
    }
    if (kpts)
        for (int i = threadIdx.x; i < 4 * K; i += 256) rec[8 + 2 * K + i] = __float_as_int(kpts[i]);
    cnt = wave_sum_i(cnt);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        rec[0] = epoch; rec[1] = n[0]; rec[2] = n[1]; rec[3] = red[0] + red[1] + red[2] + red[3];
        rec[4] = info[0]; rec[5] = 0; rec[6] = 0; rec[7] = 0;
    }
}

hipError_t launch_pack_record(const int* n, const int* matches0, const float* mscores0, const int* info, int epoch, int K,
                              int* rec, int n_pairs, const float* kpts, hipStream_t s) {
    hipLaunchKernelGGL(pack_record_kernel, dim3(n_pairs), dim3(256), 0, s, n, matches0, mscores0, info, epoch, K, rec, kpts);
    return hipGetLastError();
}

hipError_t launch_logsig(const float* z, long bstride, const LGState* st, int n_images, int n_max, float* lz, hipStream_t s) {
    hipLaunchKernelGGL(logsig_kernel, dim3((n_max + 255) / 256, n_images), dim3(256), 0, s, z, bstride, st, lz);
    return hipGetLastError();
}

}  // namespace im
