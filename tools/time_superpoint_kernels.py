#!/usr/bin/env python3
"""Per-kernel durations of one SuperPoint forward (two 1080 x 1920 gray images, the benchmark's), from the library's own HIP events
(im_profile_begin / im_profile_end), for every library named on the command line (A/B builds under build_abl/, see
tools/build_variant.sh). Each library runs in its own child process (ICEMATCH_LIB is read at import).

    python tools/time_superpoint_kernels.py [build_abl/<name>/libicematch.so ...]      (no argument: the in-tree library)
    IM_TSK_IMAGES=20 python tools/time_superpoint_kernels.py ...     the same for ten pairs per launch (times per forward of 20 images)
"""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    import torch
    from icepy4d_amd import synthetic
    from icepy4d_amd.engine import Engine
    eng = Engine(0)
    eng.load_state_dict("superpoint", synthetic.superpoint_state_dict(0))
    a, b = synthetic.stereo_pair(0, 1080, 1920)
    n_img = int(os.environ.get("IM_TSK_IMAGES", "2"))
    img = torch.from_numpy(__import__("numpy").stack([a, b])).cuda().repeat(n_img // 2, 1, 1).contiguous()
    eng.reserve(1080, 1920, n_img, 4096)
    for _ in range(4):
        eng.superpoint(img, max_kpts=4096)
    torch.cuda.synchronize()
    reps = 8
    eng.ctx.call("im_profile_begin")
    for _ in range(reps):
        eng.superpoint(img, max_kpts=4096)
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    eng.ctx.call("im_profile_end", buf, len(buf))
    prof = json.loads(buf.value.decode())
    cal = prof.pop("_empty_event_pair", None)
    ov = cal["total_ms"] / cal["count"] if cal and cal["count"] else 0.0
    out = {k: (v["total_ms"] - v["count"] * ov) / reps for k, v in prof.items()}
    print(json.dumps(out))


def main():
    libs = sys.argv[1:] or [os.path.join(ROOT, "icepy4d_amd", "csrc", "libicematch.so")]
    rows = {}
    for lib in libs:
        env = dict(os.environ, ICEMATCH_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(lib, "FAILED", r.stderr[-600:])
            continue
        rows[lib] = json.loads(line[-1])
    keys = sorted({k for v in rows.values() for k in v}, key=lambda k: -max(v.get(k, 0) for v in rows.values()))
    names = [os.path.basename(os.path.dirname(l)) for l in rows]
    print(f"{'kernel (ms per forward of 2 images)':38s}" + "".join(f"{n[:14]:>15s}" for n in names))
    for k in keys:
        print(f"{k:38s}" + "".join(f"{rows[l].get(k, float('nan')):15.4f}" for l in rows))
    print(f"{'sum':38s}" + "".join(f"{sum(rows[l].values()):15.4f}" for l in rows))


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
