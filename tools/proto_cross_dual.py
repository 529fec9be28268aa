#!/usr/bin/env python3
"""TIMING-ONLY prototype of the single-S cross attention (VERDICT r02 "Next 4"): patches a COPY of
icepy4d_amd/csrc/attention.hip (never the product source) so that the cross launch computes every S tile ONCE and carries the
instruction mix and memory traffic of serving BOTH directions from it:

  * only the blocks of image 0 of a pair work (image 1's blocks exit): 128 queries x all keys per (head, query block);
  * one key group per block, one wave per SIMD, 512 registers (the second accumulator set does not fit two waves per SIMD);
  * per 64-key step and wave, after the row direction (QK^T, softmax, PV as in the product kernel): the 32 x 64 score tile goes
    through a per-wave LDS scratch transposed (32 ds_write_b32, 8 ds_read_b128), a second online softmax runs on the transposed
    tile (column statistics = one scalar per lane again) and 64 more MFMAs accumulate O1^T [64 d][64 j] += V0^T P_col;
  * every second step the block's four partial O1 tiles are reduced through LDS (two halves of 32 KB) and 64 keys x 68 floats go
    to a partial buffer in HBM - the 71 MB per pair and layer that 16 strips of 256 rows produce at 4096 keypoints.

The OUTPUT IS NOT A CORRECT ATTENTION (operands of the column direction are stand-ins); only the duration is meaningful.
    python tools/proto_cross_dual.py build      # -> build_abl/dual/libicematch.so
    ICEMATCH_LIB=build_abl/dual/libicematch.so IM_ATTN_GROUPS=1 python tools/proto_cross_dual.py time
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def patch(src: str) -> str:
    def rep(old, new, count=1):
        nonlocal src
        assert src.count(old) >= 1, old
        src = src.replace(old, new) if count == 0 else src.replace(old, new, count)

    rep("__global__ __launch_bounds__(256 * G, 2 / G) void flash_attn_f32_kernel(AttnArgs a) {",
        "__global__ __launch_bounds__(256 * G, 1) void flash_attn_f32_kernel(AttnArgs a) {")
    rep("    if (a.active && a.active[(z >> 1) * a.pstride] == 0) return;",
        "    if (a.active && a.active[(z >> 1) * a.pstride] == 0) return;\n    const bool dual = a.cross && G == 1;\n    if (dual && (z & 1)) return;")
    # dual state + helper lambdas after the accumulators are declared
    rep("    float m_run = -INFINITY, l_run = 0.f;\n",
        """    float m_run = -INFINITY, l_run = 0.f;
    f32x16 p1a, p1b, p1c, p1d, ta, tb;               // O1^T tiles (d half, j half), transposed score tile
#pragma unroll
    for (int r = 0; r < 16; ++r) { p1a[r] = 0.f; p1b[r] = 0.f; p1c[r] = 0.f; p1d[r] = 0.f; }
    float mcol = -INFINITY, lcol = 0.f;
    float* const scr = smem + ATTN_LDS_FLOATS + ((threadIdx.x >> 6) & 3) * (64 * 36);     // per-wave transposition scratch
    float* const red = smem + ATTN_LDS_FLOATS + 4 * 64 * 36;                               // 4 waves x 32 registers x 64 lanes
    int dual_step = 0;
    auto dual_tile = [&](const f32x16& xa, const f32x16& xb) {
        const int lane_ = threadIdx.x & 63, c_ = lane_ & 31, hh_ = lane_ >> 5;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            scr[acc_row(r, hh_) * 36 + c_] = xa[r];
            scr[(32 + acc_row(r, hh_)) * 36 + c_] = xb[r];
        }
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const float4 u0 = *reinterpret_cast<const float4*>(scr + c_ * 36 + 8 * t4 + 4 * hh_);
            const float4 u1 = *reinterpret_cast<const float4*>(scr + (32 + c_) * 36 + 8 * t4 + 4 * hh_);
            ta[4 * t4] = u0.x; ta[4 * t4 + 1] = u0.y; ta[4 * t4 + 2] = u0.z; ta[4 * t4 + 3] = u0.w;
            tb[4 * t4] = u1.x; tb[4 * t4 + 1] = u1.y; tb[4 * t4 + 2] = u1.z; tb[4 * t4 + 3] = u1.w;
        }
        softmax_tile<false>(ta, tb, 0, nk, hh_, mcol, lcol, p1a, p1c);
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
            p1a = mfma32(qf[s2], ta[s2], p1a);
            p1b = mfma32(qf[s2], tb[s2], p1b);
            p1c = mfma32(qf[16 + s2], ta[s2], p1c);
            p1d = mfma32(qf[16 + s2], tb[s2], p1d);
        }
        if ((++dual_step & 1) == 0) {
            // flush: cross-wave reduction through LDS in two halves, then 64 keys x 68 floats to the partial buffer
            const int w_ = (threadIdx.x >> 6) & 3;
            // working block wb of 256, flush f of 32: distinct 17 KB slots, wrapped into the stage buffer (the product's split-KV scratch)
            float* pp = a.part + (long)(((grp_idx * 128 + head * 32 + qblk) * 32 + (dual_step >> 1)) % 3800) * (64 * 68) + lane_ * 68;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const f32x16& e0 = half ? p1b : p1a;
                const f32x16& e1 = half ? p1d : p1c;
                float4* wr = reinterpret_cast<float4*>(red) + (w_ * 8) * 64 + lane_;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    wr[g * 64] = make_float4(e0[4 * g], e0[4 * g + 1], e0[4 * g + 2], e0[4 * g + 3]);
                    wr[(4 + g) * 64] = make_float4(e1[4 * g], e1[4 * g + 1], e1[4 * g + 2], e1[4 * g + 3]);
                }
                __syncthreads();
                float4 sum0 = make_float4(0.f, 0.f, 0.f, 0.f), sum1 = sum0;
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const float4 x0 = reinterpret_cast<const float4*>(red)[(o * 8 + 2 * w_) * 64 + lane_];
                    const float4 x1 = reinterpret_cast<const float4*>(red)[(o * 8 + 2 * w_ + 1) * 64 + lane_];
                    sum0.x += x0.x; sum0.y += x0.y; sum0.z += x0.z; sum0.w += x0.w;
                    sum1.x += x1.x; sum1.y += x1.y; sum1.z += x1.z; sum1.w += x1.w;
                }
                *reinterpret_cast<float4*>(pp + half * 32 + 8 * w_) = sum0;
                *reinterpret_cast<float4*>(pp + half * 32 + 8 * w_ + 4) = sum1;
                __syncthreads();
            }
            if (w_ == 0) { pp[64] = mcol; pp[65] = lcol; }
#pragma unroll
            for (int r = 0; r < 16; ++r) { p1a[r] = 0.f; p1b[r] = 0.f; p1c[r] = 0.f; p1d[r] = 0.f; }
            mcol = -INFINITY; lcol = 0.f;
        }
    };
""")
    rep("        pv_tile(sV0 + (VRS) * VSTG + gv, c, hh, SA, SB, o0, o1);                        \\\n",
        "        pv_tile(sV0 + (VRS) * VSTG + gv, c, hh, SA, SB, o0, o1);                        \\\n        if (dual) dual_tile(SA, SB);                                                    \\\n")
    rep("        pv_tile(vr, c, hh, sa, sb, o0, o1);                                             \\\n",
        "        pv_tile(vr, c, hh, sa, sb, o0, o1);                                             \\\n        if (dual) dual_tile(sa, sb);                                                    \\\n")
    # keep the column accumulators alive to the end
    rep("    // ---- epilogue: lane (c, hh) holds query qrow",
        "    if (dual) { float keep = p1a[0] + p1b[1] + p1c[2] + p1d[3] + lcol; asm volatile(\"\" :: \"v\"(keep)); }\n    // ---- epilogue: lane (c, hh) holds query qrow")
    rep("    const size_t lds = G * ATTN_LDS_FLOATS * sizeof(float);",
        "    const size_t lds = (G * ATTN_LDS_FLOATS + (G == 1 ? 4 * 64 * 36 + 4 * 8 * 64 * 4 : 0)) * sizeof(float);")
    return src


def build():
    subprocess.run([os.path.join(ROOT, "tools", "build_variant.sh"), "dual_src_only"], check=False, capture_output=True)
    out = os.path.join(ROOT, "build_abl", "dual")
    os.makedirs(out, exist_ok=True)
    src_dir = os.path.join(ROOT, "build_abl", "dual_src_only", "src")
    text = patch(open(os.path.join(src_dir, "attention.hip")).read())
    open(os.path.join(src_dir, "attention.hip"), "w").write(text)
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                        "-Rpass-analysis=kernel-resource-usage", "-c", "attention.hip", "-o", "attention.o"], cwd=src_dir, capture_output=True, text=True)
    print("\n".join(l for l in r.stderr.splitlines() if "flash_attn" in l or "VGPRs:" in l or "Spill" in l or "error" in l)[:3000])
    assert r.returncode == 0, r.stderr[-3000:]
    objs = [f for f in os.listdir(src_dir) if f.endswith(".o")]
    subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", os.path.join(out, "libicematch.so")] + objs, cwd=src_dir, check=True)
    print("built", os.path.join(out, "libicematch.so"))


def time_it():
    sys.path.insert(0, ROOT)
    import torch
    from icepy4d_amd import _lib
    from icepy4d_amd._lib import ptr, stream_ptr
    ctx = _lib.Context(0)
    n, B = 4096, 4                                    # two pairs per launch, bench.py's default mode when this was measured
    q = torch.randn(B, 4, n, 64, device="cuda"); v = torch.randn_like(q)
    out = torch.empty(B, n, 256, device="cuda")
    dn = torch.full((B,), n, dtype=torch.int32, device="cuda")

    def run(cross):
        ctx.call("im_flash_attn", ptr(q), ptr(q), ptr(v), ptr(out), ptr(dn), n, B, 4, cross, 0.125, stream_ptr())
    for cross in (1, 0):
        for _ in range(5):
            run(cross)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run(cross)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        alg = (6.0 if cross else 8.0) * 256 * n * n * (B // 2)
        print(f"lib={os.environ.get('ICEMATCH_LIB', 'product')} groups={os.environ.get('IM_ATTN_GROUPS', '2')} cross={cross} batch={B}: {ms * 1e3:.1f} us per launch "
              f"= {ms * 1e3 / (B // 2):.1f} us per pair; algorithmic {alg / ms / 1e9:.1f} TFLOP/s = {alg / ms / 1e9 / 157.3:.3f} of the fp32-MFMA peak", flush=True)
    # the merge of the partials: 16 strips x 4096 keys x 4 heads x 68 floats per pair read once, 4 MB of output rows written
    part = torch.randn(16, (B // 2) * 4 * n * 68, device="cuda")
    for _ in range(3):
        s = part.sum(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        s = part.sum(0)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"merge stand-in (sum over 16 strip partials, {part.numel() * 4 / 1e6:.0f} MB read): {ms * 1e3:.1f} us per launch = {ms * 1e3 / (B // 2):.1f} us per pair", flush=True)


if __name__ == "__main__":
    {"build": build, "time": time_it}[sys.argv[1]]()
