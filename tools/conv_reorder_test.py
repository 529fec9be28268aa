#!/usr/bin/env python3
"""Experiment (build_abl copies only): the slab step of conv_wino.hip with the stage of slab + 1 issued AFTER the LDS reads of slab,
with optional delays, to find out why that order gave wrong results in round 3.
The kernel source is taken from git revision b135164 (the version this experiment was written against; needs the git history, i.e.
the build step runs in the development container, the built libraries travel to the GPU box).
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def kernel_source_at(rev):
    """conv_wino.hip as it was at `rev`: the patches of this experiment are text replacements against that version of the kernel."""
    import subprocess
    return subprocess.run(["git", "-C", ROOT, "show", f"{rev}:icepy4d_amd/csrc/conv_wino.hip"], check=True, capture_output=True, text=True).stdout


src0 = kernel_source_at("b135164")
a = src0.index("#define IM_SMMA(slab)")
b = src0.index("#undef IM_SSTAGE")
NEW = r'''#define IM_SREAD(slab)                                                                                  \
    {                                                                                                   \
        const float4* pa = reinterpret_cast<const float4*>(sP + ((slab) & 1) * S_SP);                   \
        if (ph == 0) {                                                                                  \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                          \
                const float4 d0 = IM_SD(0, j_), d1 = IM_SD(1, j_), d2 = IM_SD(2, j_);                   \
                t0[j_] = sub4(d0, d2); t1[j_] = add4(d1, d2);                                           \
            }                                                                                           \
        } else {                                                                                        \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                          \
                const float4 d1 = IM_SD(0, j_), d2 = IM_SD(1, j_), d3 = IM_SD(2, j_);                   \
                t0[j_] = sub4(d2, d1); t1[j_] = sub4(d1, d3);                                           \
            }                                                                                           \
        }                                                                                               \
    }
#define IM_SREADU(slab)                                                                                 \
    {                                                                                                   \
        const float4* ua = reinterpret_cast<const float4*>(sU + ((slab) & 1) * W_SU) + b_slot;          \
        _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_) u[p_] = ua[p_ * 128];                          \
    }
#define IM_SMMA()                                                                                       \
    {                                                                                                   \
        float4 v[8];                                                                                    \
        v[0] = sub4(t0[0], t0[2]); v[1] = add4(t0[1], t0[2]); v[2] = sub4(t0[2], t0[1]); v[3] = sub4(t0[1], t0[3]); \
        v[4] = sub4(t1[0], t1[2]); v[5] = add4(t1[1], t1[2]); v[6] = sub4(t1[2], t1[1]); v[7] = sub4(t1[1], t1[3]); \
        _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_) acc[p_] = mfma32(v[p_].x, u[p_].x, acc[p_]);   \
        _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_) acc[p_] = mfma32(v[p_].y, u[p_].y, acc[p_]);   \
        _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_) acc[p_] = mfma32(v[p_].z, u[p_].z, acc[p_]);   \
        _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_) acc[p_] = mfma32(v[p_].w, u[p_].w, acc[p_]);   \
    }

    const int nslab = a.Cin / WCC;
    float4 t0[4], t1[4], u[8];
    IM_SSTAGE(0)
    __syncthreads();
    for (int slab = 0; slab < nslab; ++slab) {
        ORDER
        IM_SMMA()
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
'''
ST = "if (slab + 1 < nslab) IM_SSTAGE(slab + 1)"
VARIANTS = {
    "ro_P_S_U": f"IM_SREAD(slab) {ST} IM_SREADU(slab)",
    "ro_U_S_P": f"IM_SREADU(slab) {ST} IM_SREAD(slab)",
    "ro_S_P_U": f"{ST} IM_SREAD(slab) IM_SREADU(slab)",
    "ro_P_U_S": f"IM_SREAD(slab) IM_SREADU(slab) {ST}",
    "ro_P_U_S_noasm": f"IM_SREAD(slab) IM_SREADU(slab) {ST}",
    "ro_sleep8_P_U_S": 'asm volatile("s_sleep 8" ::: "memory"); ' + f"IM_SREAD(slab) IM_SREADU(slab) {ST}",      # 512 cycles
    "ro_sleep64_P_U_S": 'asm volatile("s_sleep 64" ::: "memory"); ' + f"IM_SREAD(slab) IM_SREADU(slab) {ST}",    # 4096 cycles
    "ro_sleep127x4_P_U_S": 'asm volatile("s_sleep 127\\ns_sleep 127\\ns_sleep 127\\ns_sleep 127" ::: "memory"); ' + f"IM_SREAD(slab) IM_SREADU(slab) {ST}",
    # the same with ~400 idle cycles between the previous step's last MFMA (+ barrier) and the first LDS read
    "ro_nop_P_U_S": 'asm volatile("s_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15" ::: "memory"); ' + f"IM_SREAD(slab) IM_SREADU(slab) {ST}",
}
mode = sys.argv[1]
if mode == "build":
    procs = []
    for name, order in VARIANTS.items():
        src = src0[:a] + NEW.replace("ORDER", order) + src0[b:]
        if name.endswith("_noasm"):
            i0, i1 = src.index("__device__ __forceinline__ float4 add4("), src.index("static constexpr int S_TH = 8")
            src = src[:i0] + """__device__ __forceinline__ float4 add4(float4 x, float4 y) {
    const f32x2 a = {x.x, x.y}, b = {x.z, x.w}, c = {y.x, y.y}, d = {y.z, y.w};
    const f32x2 r0 = a + c, r1 = b + d;
    return make_float4(r0.x, r0.y, r1.x, r1.y);
}
__device__ __forceinline__ float4 sub4(float4 x, float4 y) {
    const f32x2 a = {x.x, x.y}, b = {x.z, x.w}, c = {y.x, y.y}, d = {y.z, y.w};
    const f32x2 r0 = a - c, r1 = b - d;
    return make_float4(r0.x, r0.y, r1.x, r1.y);
}

""" + src[i1:]
        src = src.replace("#undef IM_SD\n#undef IM_SMMA\n", "#undef IM_SD\n#undef IM_SREAD\n#undef IM_SREADU\n#undef IM_SMMA\n")
        out = os.path.join(ROOT, "build_abl", name)
        os.makedirs(os.path.join(out, "src"), exist_ok=True)
        for f in os.listdir(os.path.join(ROOT, "icepy4d_amd", "csrc")):
            if f.endswith(".h"):
                open(os.path.join(out, "src", f), "w").write(open(os.path.join(ROOT, "icepy4d_amd", "csrc", f)).read().replace("../../include/icematch.h", os.path.join(ROOT, "include", "icematch.h")))
        open(os.path.join(out, "src", "conv_wino.hip"), "w").write(src)
        procs.append((name, out, subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c", "conv_wino.hip", "-o", "conv_wino.o"],
                                                  cwd=os.path.join(out, "src"), stderr=subprocess.PIPE, text=True)))
    objs = [os.path.join(ROOT, "icepy4d_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "icepy4d_amd", "csrc")) if f.endswith(".o") and f != "conv_wino.o"]
    for name, out, p in procs:
        err = p.communicate()[1]
        assert p.returncode == 0, (name, err[-2000:])
        subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", os.path.join(out, "libicematch.so"), os.path.join(out, "src", "conv_wino.o")] + objs, check=True)
        print("built", name)
else:
    for name in VARIANTS:
        env = dict(os.environ, ICEMATCH_LIB=os.path.join(ROOT, "build_abl", name, "libicematch.so"))
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_kernels.py"), "-q", "-k", "winograd"], env=env, capture_output=True, text=True, cwd=ROOT)
        print(name, r.stdout.strip().splitlines()[-1], flush=True)
