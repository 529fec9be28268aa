// The two assignment sweeps (`sigmoid_log_double_softmax` + `filter_matches`, `lightglue/lightglue.py:253-306`; kernels and layout: lg_misc.hip) for
// ONE pair per launch - what a single `matcher.match()` call runs. At 4096 keypoints the 256 strips are one 4-wave block per CU, one wave per
// SIMD: the plain kernels then alternate, on every CU in step, between a phase that waits for its rows and a phase that computes on them.
#include "assign_sweep.h"

namespace im {

// The same sweep for ONE pair per launch at about one strip block per CU (4096 keypoints: 256 strips, i.e. one wave per SIMD with the whole
// register file): the next row group's loads are issued before the current group's arithmetic, from a second register buffer. Without it
// every CU alternates between a load phase and a compute phase in step with all the others and HBM idles half of the time. The same
// operations in the same order as lse_stats_kernel<true>: bit-identical (tools/bench_assign.py prints a SHA-1 of the outputs).
__global__ __launch_bounds__(256, 1) void lse_stats_pipe_kernel(AssignArgs aa) {
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
        auto load_group = [&](int g, float (&x)[4][16]) {
            load_rows16<true, 4>([&](int rr) { return sim + (long)min(i0 + 4 * g + rr, m - 1) * ld; },
                                 [&](int rr) { return 4 * g + rr < nrow ? n : 0; }, c0, lane, x);
        };
        auto group_step = [&](int g, float (&x)[4][16]) {
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
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float mx = fmaxf(fmaxf(x[0][e], x[1][e]), fmaxf(x[2][e], x[3][e]));
                const float nm = fmaxf(cM[e], mx);
                cS[e] = cS[e] * fexp(cM[e] - nm) + ((fexp(x[0][e] - nm) + fexp(x[1][e] - nm)) + (fexp(x[2][e] - nm) + fexp(x[3][e] - nm)));
                cM[e] = nm;
            }
        };
        static_assert(AS_ROWS == 16, "four row groups, two register buffers");
        float xa[4][16], xb[4][16];
        load_group(0, xa);
        load_group(1, xb);
        group_step(0, xa);
        load_group(2, xa);
        group_step(1, xb);
        load_group(3, xb);
        group_step(2, xa);
        group_step(3, xb);
        float2* cp = cpart + (long)blockIdx.x * kmax;
        const bool pvec = (kmax & 1) == 0;
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
                    if (j + e < n) cp[j + e] = make_float2(cM[4 * q + e], cS[4 * q + e]);
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

// best_sweep_kernel<MODE, true> for one pair per launch, software-pipelined like lse_stats_pipe_kernel: the rows of the next two trips are in
// flight behind the current trip's arithmetic (three register buffers in rotation). Maxima and arg-maxima only: any order gives the same bits.
template <int MODE>
__global__ __launch_bounds__(256, 1) void best_sweep_pipe_kernel(AssignArgs aa) {
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
        auto load_rows = [&](int r0, float (&x)[2][16]) {
            load_rows16<true, 2>([&](int rr) { return sim + (long)min(i0 + r0 + rr, m - 1) * ld; }, [&](int) { return n; }, c0, lane, x);
        };
        auto rows_step = [&](int r0, float (&x)[2][16]) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int r = r0 + rr;
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
                        // branch-free (selects): as `if (j < n) { if (...) {...} if (...) {...} }` every entry cost two scalar branches
                        const bool live = j < n;
                        const bool upr = live && (v > bv || bj == 0x7fffffff);      // ascending j in a lane's visit order
                        bv = upr ? v : bv;
                        bj = upr ? j : bj;
                        const bool upc = live && (v > cbv[k] || cbi[k] < 0);
                        cbv[k] = upc ? v : cbv[k];
                        cbi[k] = upc ? i0 + r : cbi[k];
                        if (e == 3) __builtin_amdgcn_sched_barrier(0);   // one quad at a time: interleaved, the 16 entries' temporaries spill
                    }
                pbv[r][tid] = bv;
                pbj[r][tid] = bj;
            }
        };
        {
            static_assert(AS_ROWS == 16, "eight trips of two rows");
            float xa[2][16], xb[2][16], xc[2][16];
            load_rows(0, xa);
            load_rows(2, xb);
            load_rows(4, xc);  rows_step(0, xa);
            load_rows(6, xa);  rows_step(2, xb);
            load_rows(8, xb);  rows_step(4, xc);
            load_rows(10, xc); rows_step(6, xa);
            load_rows(12, xa); rows_step(8, xb);
            load_rows(14, xb); rows_step(10, xc);
            rows_step(12, xa);
            rows_step(14, xb);
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

void launch_lse_stats_pipe(const AssignArgs& a, dim3 grid, hipStream_t s) { hipLaunchKernelGGL(lse_stats_pipe_kernel, grid, dim3(256), 0, s, a); }

void launch_best_sweep_pipe(const AssignArgs& a, dim3 grid, hipStream_t s) {
    if (a.mode == 0) hipLaunchKernelGGL(best_sweep_pipe_kernel<0>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(best_sweep_pipe_kernel<1>, grid, dim3(256), 0, s, a);
}

}  // namespace im
