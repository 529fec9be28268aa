#!/usr/bin/env python3
"""In-kernel cycle stamps of the Winograd kernel's slab loop (a patched COPY under build_abl/conv_stamps, never the product; the
stamp values go to a buffer of their own that nothing else reads, as /opt/skills/guides/cdna_hip_programming.md section 7 asks):
wave 0 of every block records s_memtime
    A  at the top of a slab step (after the previous step's barrier),
    B  in front of the first MFMA of the step (operand reads and input transform done),
    C  behind the last MFMA of the step (issued, not necessarily retired),
    D  behind the wait for the next slab's transfer,
    (next A) behind the barrier
and the script prints the median lengths of the four segments in cycles.
    python tools/conv_stamps.py build ; python tools/conv_stamps.py time [one_block]
The kernel source is taken from git revision 956540c (the kernel the stamps were taken on: asm LDS-DMA, before the epilogue rewrite).
"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_at(rev):
    """conv_wino.hip as it was at `rev`: the patches of this experiment are text replacements against that version of the kernel."""
    import subprocess
    return subprocess.run(["git", "-C", ROOT, "show", f"{rev}:icepy4d_amd/csrc/conv_wino.hip"], check=True, capture_output=True, text=True).stdout


OUT = os.path.join(ROOT, "build_abl", "conv_stamps")


def patch(src, one_block, aprio=False):
    def rep(old, new):
        nonlocal src
        assert src.count(old) == 1, old[:60]
        src = src.replace(old, new)
    rep("namespace im {\n\nstatic constexpr int WCC = 8;",
        "namespace im {\n\n__device__ unsigned long long im_conv_stamps[1 << 20];   // [block][slab][5]\n__device__ unsigned long long im_conv_blk[1 << 16];      // [block][start, loop begin, loop end, end, hw_id x 4]\n\nstatic constexpr int WCC = 8;")
    rep("        float4 v[8];                                                                                    \\\n",
        "        if (stamp_on) st_a = __builtin_amdgcn_s_memtime();                                              \\\n        float4 v[8];                                                                                    \\\n")
    rep("        _Pragma(\"unroll\") for (int p_ = 0; p_ < 8; ++p_) u[p_] = ua[p_ * 128];                          \\\n",
        "        _Pragma(\"unroll\") for (int p_ = 0; p_ < 8; ++p_) u[p_] = ua[p_ * 128];                          \\\n        __builtin_amdgcn_sched_barrier(0); if (stamp_on) st_b = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); \\\n")
    rep("        if constexpr (!FUSE1A) {\n            __builtin_amdgcn_sched_barrier(0);      // the MFMAs stay in FRONT of the wait and the barrier\n            IM_DMA_WAIT();",
        "        if constexpr (!FUSE1A) {\n            __builtin_amdgcn_sched_barrier(0);      // the MFMAs stay in FRONT of the wait and the barrier\n            if (stamp_on) { asm volatile(\"\" :: \"v\"(acc[7][15])); st_c = __builtin_amdgcn_s_memtime(); }\n            IM_DMA_WAIT();\n            if (stamp_on) st_d = __builtin_amdgcn_s_memtime();")
    rep("        __syncthreads();                            // FUSE1A: the compiler's own vmcnt(0) in front of the barrier covers the builtin transfers\n",
        "        __syncthreads();                            // FUSE1A: the compiler's own vmcnt(0) in front of the barrier covers the builtin transfers\n"
        "        if (stamp_on) {\n            const unsigned long long st_e = __builtin_amdgcn_s_memtime();\n"
        "            unsigned long long* sp_ = im_conv_stamps + ((long)(blockIdx.x & 8191) * 16 + slab) * 5;\n"
        "            sp_[0] = st_a; sp_[1] = st_b; sp_[2] = st_c; sp_[3] = st_d; sp_[4] = st_e;\n        }\n")
    rep("    const int nslab = a.Cin / WCC;\n",
        "    const int nslab = a.Cin / WCC;\n    const bool stamp_on = tid == 0 && !FUSE1A;\n    unsigned long long st_a = 0, st_b = 0, st_c = 0, st_d = 0;\n"
        "    if (!FUSE1A && lane == 0) im_conv_blk[(long)(blockIdx.x & 8191) * 8 + 4 + wave] = __builtin_amdgcn_s_getreg(0xf804);\n"
        "    if (stamp_on) im_conv_blk[(long)(blockIdx.x & 8191) * 8 + 1] = __builtin_amdgcn_s_memtime();\n")
    rep("    if (rtile >= ntile) return;\n", "    if (rtile >= ntile) return;\n    const unsigned long long st_start = __builtin_amdgcn_s_memtime();\n")
    rep("    // ---- inverse transform Y = A^T M A.", "    if (stamp_on) { im_conv_blk[(long)(blockIdx.x & 8191) * 8 + 0] = st_start; im_conv_blk[(long)(blockIdx.x & 8191) * 8 + 2] = __builtin_amdgcn_s_memtime(); }\n    // ---- inverse transform Y = A^T M A.")
    rep("    if (ph == 0) finish(std::integral_constant<int, 0>{});\n    else finish(std::integral_constant<int, 1>{});\n",
        "    if (ph == 0) finish(std::integral_constant<int, 0>{});\n    else finish(std::integral_constant<int, 1>{});\n    if (stamp_on) { asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); im_conv_blk[(long)(blockIdx.x & 8191) * 8 + 3] = __builtin_amdgcn_s_memtime(); }\n")
    src += "\nextern \"C\" int im_debug_conv_stamps(unsigned long long* h_dst, size_t n) {\n    return (int)hipMemcpyFromSymbol(h_dst, HIP_SYMBOL(im::im_conv_stamps), n * sizeof(unsigned long long));\n}\n"
    src += "extern \"C\" int im_debug_conv_blk(unsigned long long* h_dst, size_t n) {\n    return (int)hipMemcpyFromSymbol(h_dst, HIP_SYMBOL(im::im_conv_blk), n * sizeof(unsigned long long));\n}\n"
    if aprio:
        rep("    const unsigned long long st_start = __builtin_amdgcn_s_memtime();\n", "    const unsigned long long st_start = __builtin_amdgcn_s_memtime();\n    if (__builtin_amdgcn_s_getreg(0x1804) & 1) __builtin_amdgcn_s_setprio(3);\n")
    if one_block:
        rep("    const size_t lds = (S_LDS_FLOATS + (FUSE ? S_FUSE : 0)) * sizeof(float);",
            "    const size_t lds = (S_LDS_FLOATS + (FUSE ? S_FUSE : 0)) * sizeof(float) + 16384;")
    return src


def build():
    for name, ob, ap in (("conv_stamps", False, False), ("conv_stamps_1blk", True, False), ("conv_stamps_aprio", False, True)):
        out = os.path.join(ROOT, "build_abl", name)
        os.makedirs(os.path.join(out, "src"), exist_ok=True)
        for f in os.listdir(os.path.join(ROOT, "icepy4d_amd", "csrc")):
            if f.endswith(".h"):
                open(os.path.join(out, "src", f), "w").write(open(os.path.join(ROOT, "icepy4d_amd", "csrc", f)).read().replace("../../include/icematch.h", os.path.join(ROOT, "include", "icematch.h")))
        open(os.path.join(out, "src", "conv_wino.hip"), "w").write(patch(kernel_source_at("956540c"), ob, ap))
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c", "conv_wino.hip", "-o", "conv_wino.o"], cwd=os.path.join(out, "src"), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        objs = [os.path.join(ROOT, "icepy4d_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "icepy4d_amd", "csrc")) if f.endswith(".o") and f != "conv_wino.o"]
        subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", os.path.join(out, "libicematch.so"), os.path.join(out, "src", "conv_wino.o")] + objs, check=True)
        print("built", name)


def time_it():
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from icepy4d_amd import _lib
    from icepy4d_amd._lib import ptr, stream_ptr
    ctx = _lib.Context(0)
    for (h, w, cin, cout, pool) in ((1080, 1920, 64, 64, 1), (270, 480, 128, 128, 1)):
        x = torch.randn(2, h, w, cin, device="cuda"); wt = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5; b = torch.randn(cout)
        out = torch.empty(2, h // 2 if pool else h, w // 2 if pool else w, cout, device="cuda")
        for _ in range(3):
            ctx.call("im_conv3x3_winograd", ptr(x), ptr(wt), ptr(b), ptr(out), 2, h, w, cin, cout, 1, pool, stream_ptr())
        torch.cuda.synchronize()
        nslab = cin // 8
        buf = np.zeros(8192 * 16 * 5, dtype=np.uint64)
        fn = ctx.lib.im_debug_conv_stamps
        fn.argtypes = [C.c_void_p, C.c_size_t]
        assert fn(buf.ctypes.data, buf.size) == 0
        s = buf.reshape(8192, 16, 5)[:, :nslab].astype(np.int64)
        s = s[s[:, 0, 0] > 0]
        ab, bc, cd, de = s[..., 1] - s[..., 0], s[..., 2] - s[..., 1], s[..., 3] - s[..., 2], s[..., 4] - s[..., 3]
        step = s[:, 1:, 0] - s[:, :-1, 0]
        med = lambda a: int(np.median(a))
        blk = np.zeros(8192 * 8, dtype=np.uint64)
        fb = ctx.lib.im_debug_conv_blk
        fb.argtypes = [C.c_void_p, C.c_size_t]
        assert fb(blk.ctypes.data, blk.size) == 0
        bk = blk.reshape(8192, 8).astype(np.int64)
        bk = bk[bk[:, 0] > 0]
        hw = bk[:, 4:8]
        # HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh [12], se [15:13]... 
        simd = (hw >> 4) & 3; slot = hw & 15; cu = (hw >> 8) & 15; se = (hw >> 13) & 7
        import collections
        print("   HW_ID fields over all blocks: wave_id", dict(collections.Counter(slot.ravel().tolist())), " tg_id", dict(collections.Counter(((hw >> 16) & 15).ravel().tolist())),
              " queue", dict(collections.Counter(((hw >> 24) & 7).ravel().tolist())), " bits[31:27]", dict(collections.Counter((hw >> 27).ravel().tolist())))
        for wid in (0, 1):
            selb = slot[:, 0] == wid
            print(f"   wave_id {wid}: blocks {int(selb.sum())}, lifetime {med((bk[:, 3] - bk[:, 0])[selb])}, loop {med((bk[:, 2] - bk[:, 1])[selb])}, epilogue {med((bk[:, 3] - bk[:, 2])[selb])}")
        print("   block lifetime", med(bk[:, 3] - bk[:, 0]), "prologue", med(bk[:, 1] - bk[:, 0]), "loop", med(bk[:, 2] - bk[:, 1]), "epilogue", med(bk[:, 3] - bk[:, 2]),
              "| SIMD ids of waves 0-3 (first blocks):", simd[:6].tolist(), "slots:", slot[:6].tolist(), "distinct SIMDs per block (mean):", float(np.mean([len(set(r)) for r in simd.tolist()])))
        print(f"{os.environ.get('ICEMATCH_LIB', 'product')} conv {h}x{w} {cin}->{cout}: blocks {len(s)}; step {med(step)} cycles (p10 {int(np.percentile(step, 10))}, p90 {int(np.percentile(step, 90))}); "
              f"reads+transform {med(ab)}, MFMA issue {med(bc)}, transfer wait {med(cd)}, barrier {med(de)}; A(next) - E {med(s[:, 1:, 0] - s[:, :-1, 4])}", flush=True)


if __name__ == "__main__":
    {"build": build, "time": time_it}[sys.argv[1]]()
