// Shared by the assignment sweeps of lg_misc.hip and their one-pair-per-launch forms in lg_assign_pipe.hip (kept in a translation unit
// of their own: with them next to the plain kernels the compiler's inlining / register allocation of the plain kernels changed - 122 ->
// 128 registers and spills in lse_stats_kernel<true> - although their source had not).
#pragma once
#include "common.h"
#include "lg_misc.h"

namespace im {

static constexpr int AS_ROWS = 16;       // rows per block (= per strip of the column partials)
static constexpr int AS_CHUNK = 1024;    // columns a wave takes per pass: 4 x (64 lanes x float4)
static constexpr float AS_NEG = -3.0e38f;  // stands in for -inf on masked entries (finite: no inf - inf in the online merges)

// the arguments of pair `pr` of a batch (blockIdx.y of every assignment kernel): every buffer is laid out [pair][...]
__device__ __forceinline__ AssignArgs for_pair(AssignArgs a, int pr) {
    a.sim += (long)pr * a.sim_ps;
    a.m_ptr += (long)pr * a.state_ps; a.n_ptr += (long)pr * a.state_ps;
    if (a.lz0) { a.lz0 += (long)pr * a.lz_ps; a.lz1 += (long)pr * a.lz_ps; }
    a.rmax += (long)pr * a.vec_ps; a.rlog += (long)pr * a.vec_ps; a.cmax += (long)pr * a.vec_ps; a.clog += (long)pr * a.vec_ps;
    a.ridx += (long)pr * a.vec_ps; a.rval += (long)pr * a.vec_ps; a.cbest += (long)pr * a.vec_ps;
    a.part += (long)pr * a.part_ps;
    if (a.ind0) { a.ind0 += (long)pr * a.out_ps; a.ind1 += (long)pr * a.out_ps; }
    a.out_m0 += (long)pr * a.out_ps; a.out_m1 += (long)pr * a.out_ps; a.out_s0 += (long)pr * a.out_ps; a.out_s1 += (long)pr * a.out_ps;
    return a;
}

// 16 values of row `p` (chunk base c0): columns c0 + q * 256 + lane * 4 + e; entries >= n read as AS_NEG. Branch-free: a quad that
// starts past the row's end is loaded from the row's last quad instead and masked (n = 0 masks the whole row: callers pass that for
// rows past the strip's end). With the loads behind `if (j < n)` every load was a basic block of its own, the compiler collected the
// loads of all 16 rows of a strip at the top of the chunk loop, and the sweeps needed 284-320 registers: one wave per SIMD.
template <bool VEC>
__device__ __forceinline__ void load_row16(const float* __restrict__ p, int c0, int lane, int n, float (&x)[16]) {
    const int last = max(n - 1, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = c0 + q * 256 + lane * 4;
        if constexpr (VEC) {
            const float4 v = *reinterpret_cast<const float4*>(p + min(j, last & ~3));   // ld % 4 == 0: the quad lies inside the row's storage
            x[4 * q] = (j < n) ? v.x : AS_NEG; x[4 * q + 1] = (j + 1 < n) ? v.y : AS_NEG;
            x[4 * q + 2] = (j + 2 < n) ? v.z : AS_NEG; x[4 * q + 3] = (j + 3 < n) ? v.w : AS_NEG;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[4 * q + e] = (j + e < n) ? p[min(j + e, last)] : AS_NEG;
        }
    }
}

// R rows at once, ALL their 16-byte loads issued before the first of them is consumed. Written as two phases with a scheduling fence in
// between: with the masking selects next to each load (load_row16 row by row) the compiler, holding the kernels at 122 / 155 registers,
// reused one 4-register temporary for every load and put `s_waitcnt vmcnt(0)` behind each - ONE kilobyte in flight per wave, the sweeps
// at 1.7 TB/s with 71 % of the wave cycles waiting (round 5, `SQ_WAIT_ANY`). rowp(rr) / rown(rr): row pointer and live length of row rr.
template <bool VEC, int R, typename RowP, typename RowN>
__device__ __forceinline__ void load_rows16(RowP rowp, RowN rown, int c0, int lane, float (&x)[R][16]) {
    if constexpr (VEC) {
        float4 raw[R][4];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const float* __restrict__ p = rowp(rr);
            const int last = max(rown(rr) - 1, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) raw[rr][q] = *reinterpret_cast<const float4*>(p + min(c0 + q * 256 + lane * 4, last & ~3));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int n = rown(rr);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = c0 + q * 256 + lane * 4;
                const float4 v = raw[rr][q];
                x[rr][4 * q] = (j < n) ? v.x : AS_NEG; x[rr][4 * q + 1] = (j + 1 < n) ? v.y : AS_NEG;
                x[rr][4 * q + 2] = (j + 2 < n) ? v.z : AS_NEG; x[rr][4 * q + 3] = (j + 3 < n) ? v.w : AS_NEG;
            }
        }
    } else {
#pragma unroll
        for (int rr = 0; rr < R; ++rr) load_row16<false>(rowp(rr), c0, lane, rown(rr), x[rr]);
    }
}

// exp of a non-positive difference to a running maximum: v_exp_f32 on x * log2(e) (2 instructions instead of expf's ~12; the
// sweeps below evaluate 2.3 of them per matrix entry). Relative error <= 2^-22 for |x| < 16, growing with |x| * 2^-24.
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// MODE 0 (LightGlue): score = log_softmax_row + log_softmax_col + certainties
// MODE 1 (SuperGlue, `superglue.py:160, 185`): score = ((x + u_i) + v_j) - norm, with u in rmax, v in cmax, norm in *rlog
template <int MODE>
__device__ __forceinline__ float assign_score(float x, float rm, float rl, float cm, float cl, float l0, float l1) {
    if constexpr (MODE == 0) return (((x - rm) - rl) + ((x - cm) - cl)) + (l0 + l1);
    else return ((x + rm) + cm) - rl;
}


void launch_lse_stats_pipe(const AssignArgs& a, dim3 grid, hipStream_t s);
void launch_best_sweep_pipe(const AssignArgs& a, dim3 grid, hipStream_t s);

}  // namespace im
