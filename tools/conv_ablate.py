#!/usr/bin/env python3
"""Timing ablations of the Winograd convolution kernel (copies of csrc/conv_wino.hip patched under build_abl/, results are WRONG by
construction, only durations mean anything): which part of a slab step / of the block costs how much.
    python tools/conv_ablate.py build          # build_abl/cv_<name>/libicematch.so for every ablation
    python tools/conv_ablate.py time           # on the GPU: times every variant on the SuperPoint shapes
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ABL = {
    "base": [],
    # no U transfer: the 8 LDS-DMA pieces of the U block per slab and wave are not issued (the patch still is)
    "no_udma": [("            else dma16(ruw, ub + k_ * 4096u, uv, so_ + k_ * u_k_bytes);                                 \\",
                 "            else if (so_ == 0xFFFFFFFFu) dma16(ruw, ub + k_ * 4096u, uv, so_ + k_ * u_k_bytes);           \\")],
    # no patch transfer (non-fused layers): the two patch pieces per slab are not issued
    "no_pdma": [("                dma16(rin, pb_, pv, (slab) * (WCC * 4u));                                               \\",
                 "                if (pv == 0x12345u) dma16(rin, pb_, pv, (slab) * (WCC * 4u));                           \\"),
                ("                dma16(rin, pb_ + S_QUAD * 16u, pv, (slab) * (WCC * 4u) + 16u);                          \\",
                 "                if (pv == 0x12345u) dma16(rin, pb_ + S_QUAD * 16u, pv, (slab) * (WCC * 4u) + 16u);      \\")],
    # one block per CU instead of two (LDS request padded past half of the 160 KB)
    "one_block": [("    const size_t lds = (S_LDS_FLOATS + (FUSE ? S_FUSE : 0)) * sizeof(float);",
                   "    const size_t lds = (S_LDS_FLOATS + (FUSE ? S_FUSE : 0)) * sizeof(float) + 16384;")],
    # no input transform: the packed adds are dropped (the MFMA operands are whatever two patch reads hold)
    "no_xform": [("t0[j_] = sub4(d0, d2, m1); t1[j_] = add4(d1, d2);", "t0[j_] = IM_SD(0, 0); t1[j_] = IM_SD(1, 0);"),
                 ("t0[j_] = sub4(d2, d1, m1); t1[j_] = sub4(d1, d3, m1);", "t0[j_] = IM_SD(0, 0); t1[j_] = IM_SD(1, 0);"),
                 ("v[0] = sub4(t0[0], t0[2], m1); v[1] = add4(t0[1], t0[2]); v[2] = sub4(t0[2], t0[1], m1); v[3] = sub4(t0[1], t0[3], m1);",
                  "v[0] = t0[0]; v[1] = t0[1]; v[2] = t0[2]; v[3] = t0[3];"),
                 ("v[4] = sub4(t1[0], t1[2], m1); v[5] = add4(t1[1], t1[2]); v[6] = sub4(t1[2], t1[1], m1); v[7] = sub4(t1[1], t1[3], m1);",
                  "v[4] = t1[0]; v[5] = t1[1]; v[6] = t1[2]; v[7] = t1[3];")],
    # one U read per slab instead of eight
    "no_uread": [("        _Pragma(\"unroll\") for (int p_ = 0; p_ < 8; ++p_) u[p_] = ua[p_ * 128];                          \\",
                  "        _Pragma(\"unroll\") for (int p_ = 0; p_ < 8; ++p_) u[p_] = ua[0];                                 \\")],
    # no MFMA: operands are consumed by a cheap asm so that the loads and the transform stay
    "no_mfma": [("acc[p_] = mfma32(v[p_].x, u[p_].x, (FIRST) ? f32x16{} : acc[p_]);",
                 "{ if (FIRST) acc[p_] = f32x16{}; asm volatile(\"\" :: \"v\"(v[p_].x), \"v\"(v[p_].y), \"v\"(v[p_].z), \"v\"(v[p_].w), \"v\"(u[p_].x), \"v\"(u[p_].y), \"v\"(u[p_].z), \"v\"(u[p_].w)); }"),
                ("acc[p_] = mfma32(v[p_].y, u[p_].y, acc[p_]);", ";"), ("acc[p_] = mfma32(v[p_].z, u[p_].z, acc[p_]);", ";"),
                ("acc[p_] = mfma32(v[p_].w, u[p_].w, acc[p_]);", ";")],
    # no epilogue: the accumulators are consumed, nothing is exchanged or stored
    "no_epi": [("    const f32x16 sa0 = (acc[0] + acc[1]) + acc[2],",
                "    { float keep_ = 0.f; _Pragma(\"unroll\") for (int p_ = 0; p_ < 8; ++p_) keep_ += acc[p_][0] + acc[p_][15]; if (keep_ == 12345.678f) a.out[0] = keep_; return; }\n    const f32x16 sa0 = (acc[0] + acc[1]) + acc[2],")],
    # epilogue without the LDS exchange between the two V-row halves (own partials used twice, no barrier)
    "no_xchg": [("        __syncthreads();\n        const float4* xr = reinterpret_cast<const float4*>(sX) + ((P ^ 1) * 8) * 128 + (tid & 127);",
                 "        const float4* xr = reinterpret_cast<const float4*>(sX) + (P * 8) * 128 + (tid & 127);")],
    # main loop only one slab (prologue + 1 step + epilogue)
    "one_slab": [("    const int nslab = a.Cin / WCC;", "    const int nslab = 1;")],
}


def build():
    procs = []
    for name, edits in ABL.items():
        out = os.path.join(ROOT, "build_abl", "cv_" + name)
        os.makedirs(os.path.join(out, "src"), exist_ok=True)
        src = open(os.path.join(ROOT, "icepy4d_amd", "csrc", "conv_wino.hip")).read()
        for old, new in edits:
            assert src.count(old) >= 1, (name, old[:60])
            src = src.replace(old, new)
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


def time_all():
    for name in ABL:
        env = dict(os.environ, ICEMATCH_LIB=os.path.join(ROOT, "build_abl", "cv_" + name, "libicematch.so"))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_kernels.py"), "conv"], env=env, capture_output=True, text=True)
        vals = []
        for line in r.stdout.splitlines():
            if line.startswith("conv "):
                shape = line.split(":")[0]
                ms = line.split("conv3x3_winograd ")[1].split(" ms")[0]
                vals.append(f"{shape.replace('conv ', '')}={ms}")
        print(f"{name:9s}", "  ".join(vals), flush=True)


if __name__ == "__main__":
    {"build": build, "time": time_all}[sys.argv[1]]()
