#!/usr/bin/env python3
"""Experiment (build_abl copy): conv_wino.hip's slab step with the LDS operand reads written as inline asm, so that the compiler
does not order them behind the LDS-DMA of the next slab with `s_waitcnt vmcnt(0)`: the transfer issued at the top of a step stays
in flight during the step's reads, transform and MFMAs and is waited for at the barrier only.
The kernel source is taken from git revision b135164 (the version this experiment was written against).
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def kernel_source_at(rev):
    """conv_wino.hip as it was at `rev`: the patches of this experiment are text replacements against that version of the kernel."""
    import subprocess
    return subprocess.run(["git", "-C", ROOT, "show", f"{rev}:icepy4d_amd/csrc/conv_wino.hip"], check=True, capture_output=True, text=True).stdout


src = kernel_source_at("b135164")
a = src.index("#define IM_SD(i, j) pa[a_slot")
b = src.index("    const int nslab = a.Cin / WCC;")
NEW = r'''    // LDS byte addresses of this lane's operand slots in stage 0 (stage 1: + S_SP * 4 resp. + W_SU * 4)
    const unsigned pa_addr = (unsigned)(unsigned long)(lds_ptr_t)(sP) + (unsigned)a_slot * 16u;
    const unsigned ua_addr = (unsigned)(unsigned long)(lds_ptr_t)(sU) + (unsigned)b_slot * 16u;
#define IM_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define IM_SOFF(i, j) (16 * ((i) * S_ROW + ((j) & 1) * S_PAR + ((j) >> 1)))
#define IM_SMMA(slab)                                                                                   \
    {                                                                                                   \
        const unsigned pa_ = pa_addr + ((slab) & 1) * (S_SP * 4u), ua_ = ua_addr + ((slab) & 1) * (W_SU * 4u); \
        f32x4 d00, d01, d02, d10, d11, d12, d20, d21, d22, d30, d31, d32, u0, u1, u2, u3, u4, u5, u6, u7; \
        IM_RD(d00, pa_, IM_SOFF(0, 0)); IM_RD(d01, pa_, IM_SOFF(1, 0)); IM_RD(d02, pa_, IM_SOFF(2, 0)); \
        IM_RD(d10, pa_, IM_SOFF(0, 1)); IM_RD(d11, pa_, IM_SOFF(1, 1)); IM_RD(d12, pa_, IM_SOFF(2, 1)); \
        IM_RD(d20, pa_, IM_SOFF(0, 2)); IM_RD(d21, pa_, IM_SOFF(1, 2)); IM_RD(d22, pa_, IM_SOFF(2, 2)); \
        IM_RD(d30, pa_, IM_SOFF(0, 3)); IM_RD(d31, pa_, IM_SOFF(1, 3)); IM_RD(d32, pa_, IM_SOFF(2, 3)); \
        IM_RD(u0, ua_, 0); IM_RD(u1, ua_, 2048); IM_RD(u2, ua_, 4096); IM_RD(u3, ua_, 6144);           \
        IM_RD(u4, ua_, 8192); IM_RD(u5, ua_, 10240); IM_RD(u6, ua_, 12288); IM_RD(u7, ua_, 14336);     \
        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(d00), "+v"(d01), "+v"(d02), "+v"(d10), "+v"(d11), "+v"(d12), "+v"(d20), "+v"(d21), "+v"(d22), "+v"(d30), "+v"(d31), "+v"(d32)); \
        float4 t0[4], t1[4], v[8];                                                                      \
        const float4 dd[4][3] = {{F4(d00), F4(d01), F4(d02)}, {F4(d10), F4(d11), F4(d12)}, {F4(d20), F4(d21), F4(d22)}, {F4(d30), F4(d31), F4(d32)}}; \
        if (ph == 0) {                                                                                  \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) { t0[j_] = sub4(dd[j_][0], dd[j_][2]); t1[j_] = add4(dd[j_][1], dd[j_][2]); } \
        } else {                                                                                        \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) { t0[j_] = sub4(dd[j_][1], dd[j_][0]); t1[j_] = sub4(dd[j_][0], dd[j_][2]); } \
        }                                                                                               \
        v[0] = sub4(t0[0], t0[2]); v[1] = add4(t0[1], t0[2]); v[2] = sub4(t0[2], t0[1]); v[3] = sub4(t0[1], t0[3]); \
        v[4] = sub4(t1[0], t1[2]); v[5] = add4(t1[1], t1[2]); v[6] = sub4(t1[2], t1[1]); v[7] = sub4(t1[1], t1[3]); \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)); \
        const float4 u[8] = {F4(u0), F4(u1), F4(u2), F4(u3), F4(u4), F4(u5), F4(u6), F4(u7)};         \
        _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_) acc[p_] = mfma32(v[p_].x, u[p_].x, acc[p_]);   \
        _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_) acc[p_] = mfma32(v[p_].y, u[p_].y, acc[p_]);   \
        _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_) acc[p_] = mfma32(v[p_].z, u[p_].z, acc[p_]);   \
        _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_) acc[p_] = mfma32(v[p_].w, u[p_].w, acc[p_]);   \
    }
#define F4(x) make_float4((x)[0], (x)[1], (x)[2], (x)[3])

'''
src2 = src[:a] + NEW + src[b:]
src2 = src2.replace("#undef IM_SD\n#undef IM_SMMA\n", "#undef IM_SMMA\n#undef IM_RD\n#undef IM_SOFF\n#undef F4\n")
# ph == 1 reads patch rows 1, 2, 3: the original IM_SD(0..2) is relative to a_slot which already includes ph, so the same offsets serve both
# the MFMAs must stay in front of the barrier
src2 = src2.replace("        IM_SMMA(slab)\n        __syncthreads();", "        IM_SMMA(slab)\n        __builtin_amdgcn_sched_barrier(0);\n        __syncthreads();")
mode = sys.argv[1]
out = os.path.join(ROOT, "build_abl", "cv_asmread")
if mode == "build":
    os.makedirs(os.path.join(out, "src"), exist_ok=True)
    for f in os.listdir(os.path.join(ROOT, "icepy4d_amd", "csrc")):
        if f.endswith(".h"):
            open(os.path.join(out, "src", f), "w").write(open(os.path.join(ROOT, "icepy4d_amd", "csrc", f)).read().replace("../../include/icematch.h", os.path.join(ROOT, "include", "icematch.h")))
    open(os.path.join(out, "src", "conv_wino.hip"), "w").write(src2)
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage", "-c", "conv_wino.hip", "-o", "conv_wino.o"],
                       cwd=os.path.join(out, "src"), capture_output=True, text=True)
    print("\n".join(l for l in r.stderr.splitlines() if "VGPRs:" in l or "Spill" in l or "error" in l)[:3000])
    assert r.returncode == 0, r.stderr[-3000:]
    objs = [os.path.join(ROOT, "icepy4d_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "icepy4d_amd", "csrc")) if f.endswith(".o") and f != "conv_wino.o"]
    subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", os.path.join(out, "libicematch.so"), os.path.join(out, "src", "conv_wino.o")] + objs, check=True)
    print("built")
else:
    env = dict(os.environ, ICEMATCH_LIB=os.path.join(out, "libicematch.so"))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_kernels.py"), "-q", "-k", "conv or superpoint"], env=env, capture_output=True, text=True, cwd=ROOT)
    print(r.stdout.strip().splitlines()[-1])
    for lib in (env["ICEMATCH_LIB"], None):
        e = dict(os.environ)
        if lib:
            e["ICEMATCH_LIB"] = lib
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_kernels.py"), "conv"], env=e, capture_output=True, text=True)
        print("asmread" if lib else "product", "  ".join(l.split(":")[0].replace("conv ", "") + "=" + l.split("conv3x3_winograd ")[1].split(" ms")[0] for l in r.stdout.splitlines() if l.startswith("conv ")))
