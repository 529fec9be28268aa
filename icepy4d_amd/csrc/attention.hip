// Flash-style fp32 attention on the f32-input matrix cores: the K x K (or M x N) attention matrix is never
// materialised (the reference materialises it on CPU: `lightglue/lightglue.py:120-123` via SDPA,
// `:200-210` explicit einsum/softmax, `SuperGlue/models/superglue.py:87-93`).
//
//   out[z][i][h*64 + d] = sum_j softmax_j(scale * q_z[h][i] . k_y[h][j]) * v_y[h][j][d],   y = cross ? z^1 : z
//
// With cross = 1 and q = k = the shared `to_qk` projection this is exactly LightGlue's bidirectional
// cross attention: softmax over the rows of sim for image 0, over its columns for image 1.
//
// Work split: one wave owns 32 query rows, a block is 4 waves = 128 queries of one (image, head); all four
// waves stream the same 64-key K/V tiles through LDS. Both products keep the QUERY on the MFMA lane:
//     S^T (keys x queries)  = K . Q^T     A = K tile from LDS (16-byte reads, permuted d order), B = Q in registers
//     O^T (d x queries)    += V^T . P^T   A = V tile from LDS, B = P = exp(S^T - m) straight from the accumulator layout
// so the online-softmax statistics (running max m, running sum l) are one scalar per lane and rescaling the
// output accumulator is a per-lane multiply. fp32 throughout; exp via v_exp_f32.
#include "common.h"
#include "kernels.h"

namespace im {

static constexpr int KT = 64;       // keys per LDS tile
static constexpr int KS = 68;       // K tile row stride (floats): conflict-free ds_read_b128
static constexpr int VS = 64;       // V tile row stride

__global__ __launch_bounds__(256) void flash_attn_f32_kernel(AttnArgs a) {
    if (a.active && *a.active == 0) return;
    __shared__ __attribute__((aligned(16))) float smem[KT * KS + KT * VS];
    float* sK = smem;
    float* sV = smem + KT * KS;

    const int z = blockIdx.z, head = blockIdx.y;
    const int y = a.cross ? (z ^ 1) : z;
    const int nq = a.n_ptr ? a.n_ptr[z] : a.n_max;
    const int nk = a.n_ptr ? a.n_ptr[y] : a.n_max;
    const int qb = blockIdx.x * 128;
    if (qb >= nq || nk <= 0) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const int qrow = qb + wave * 32 + c;
    const int qrow_ld = min(qrow, nq - 1);

    const float* Q = a.q + (long)z * a.bstride + (long)head * a.hstride;
    const float* K = a.k + (long)y * a.bstride + (long)head * a.hstride;
    const float* V = a.v + (long)y * a.bstride + (long)head * a.hstride;

    // Q fragment: lane (c, hh) keeps Q[qrow][32*hh + s], s = 0..31
    float qf[32];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        float4 v4 = *reinterpret_cast<const float4*>(Q + (long)qrow_ld * 64 + hh * 32 + t * 4);
        qf[4 * t + 0] = v4.x; qf[4 * t + 1] = v4.y; qf[4 * t + 2] = v4.z; qf[4 * t + 3] = v4.w;
    }

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    // register staging of the next K/V tile: 64 rows x 16 float4 = 1024 float4 per tensor, 4 per thread
    float4 rk[4], rv[4];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            int idx = tid + it * 256;
            int row = kt + (idx >> 4), c4 = idx & 15;
            if (row < nk) {
                rk[it] = *reinterpret_cast<const float4*>(K + (long)row * 64 + c4 * 4);
                rv[it] = *reinterpret_cast<const float4*>(V + (long)row * 64 + c4 * 4);
            } else {
                rk[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                rv[it] = rk[it];
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            int idx = tid + it * 256;
            int row = idx >> 4, c4 = idx & 15;
            *reinterpret_cast<float4*>(sK + row * KS + c4 * 4) = rk[it];
            *reinterpret_cast<float4*>(sV + row * VS + c4 * 4) = rv[it];
        }
    };

    load_tile(0);
    for (int kt = 0; kt < nk; kt += KT) {
        __syncthreads();  // previous tile fully consumed
        store_tile();
        __syncthreads();
        if (kt + KT < nk) load_tile(kt + KT);

#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int kb = kt + sub * 32;
            if (kb >= nk) break;  // wave-uniform
            // ---- S^T = K . Q^T : 32 MFMAs, contraction order d = s (hh=0) / 32+s (hh=1)
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
            const float* kp = sK + (sub * 32 + c) * KS + hh * 32;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                float4 kf = *reinterpret_cast<const float4*>(kp + t * 4);
                st = mfma32(kf.x, qf[4 * t + 0], st);
                st = mfma32(kf.y, qf[4 * t + 1], st);
                st = mfma32(kf.z, qf[4 * t + 2], st);
                st = mfma32(kf.w, qf[4 * t + 3], st);
            }
            // ---- online softmax over the 32 keys of this sub-tile (16 here, 16 in lane ^ 32)
            const bool tail = kb + 32 > nk;
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sv = st[r] * a.scale;
                if (tail && (kb + acc_row(r, hh) >= nk)) sv = -INFINITY;
                st[r] = sv;
                mx = fmaxf(mx, sv);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __expf(m_run - m_new);
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = __expf(st[r] - m_new);
                st[r] = p;
                rs += p;
            }
            rs += __shfl_xor(rs, 32);
            l_run = l_run * alpha + rs;
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            // ---- O^T += V^T . P^T : register r of P is key acc_row(r, hh) of the sub-tile
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* vp = sV + (sub * 32 + acc_row(r, hh)) * VS + c;
                o0 = mfma32(vp[0], st[r], o0);
                o1 = mfma32(vp[32], st[r], o1);
            }
        }
    }

    // ---- epilogue: lane (c, hh) holds query qrow, d = db*32 + 8*g + 4*hh + (0..3) in registers 4g..4g+3
    if (qrow < nq) {
        const float inv = 1.f / l_run;
        float* op = a.out + (long)z * a.out_bstride + (long)qrow * a.ldo + head * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 w0 = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            float4 w1 = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
            *reinterpret_cast<float4*>(op + 8 * g + 4 * hh) = w0;
            *reinterpret_cast<float4*>(op + 32 + 8 * g + 4 * hh) = w1;
        }
    }
}

hipError_t launch_flash_attn(const AttnArgs& a, hipStream_t s) {
    if (a.n_max <= 0) return hipSuccess;
    dim3 grid((a.n_max + 127) / 128, a.heads, a.batch), block(256);
    hipLaunchKernelGGL(flash_attn_f32_kernel, grid, block, 0, s, a);
    return hipGetLastError();
}

}  // namespace im
