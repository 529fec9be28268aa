#!/usr/bin/env python3
"""simple_nms on two 1080 x 1920 score maps (the two images of a benchmark pair): the stage entry point `im_nms` alone (HIP events over
`reps` calls) and the `nms_select` launch class of a SuperPoint forward (NMS + candidate epilogue + top-k selection, the library's own
events), for the kernel forms the library can be switched to (environment read once per process: one child process per form), with SHA-1s of the
map and of the selected keypoints to compare them bit for bit. (Round 5 measured the single-launch kernel of rounds 2-4 against the three-launch
form with it: 40.4 / 26.5 us; that kernel now lives in tools/experiments/nms_fused_single_launch.hip.txt.)

    python tools/bench_nms.py [reps=50]
"""
import ctypes
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FORMS = [("three launches, one per round, bit masks between them (default)", {}), ("five launches, byte masks (IM_NMS_STAGED=1; the path for r > 4)", {"IM_NMS_STAGED": "1"})]


def child(reps):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from icepy4d_amd import synthetic
    from icepy4d_amd._lib import ptr
    from icepy4d_amd.engine import Engine
    eng = Engine(0)
    eng.load_state_dict("superpoint", synthetic.superpoint_state_dict(0))
    a, b = synthetic.stereo_pair(0, 1080, 1920)
    img = torch.from_numpy(np.stack([a, b])).cuda()
    eng.reserve(1080, 1920, 2, 4096)
    res = {}
    for B in (2, 20):
        pairs = img.repeat(B // 2, 1, 1).contiguous()
        eng.reserve(1080, 1920, B, 4096)
        for _ in range(3):
            eng.superpoint(pairs, max_kpts=4096)
        torch.cuda.synchronize()
        eng.ctx.call("im_profile_begin")
        for _ in range(8):
            eng.superpoint(pairs, max_kpts=4096)
        torch.cuda.synchronize()
        buf = ctypes.create_string_buffer(1 << 16)
        eng.ctx.call("im_profile_end", buf, len(buf))
        prof = json.loads(buf.value.decode())
        cal = prof.pop("_empty_event_pair", None)
        ov = cal["total_ms"] / cal["count"] if cal and cal["count"] else 0.0
        res[f"nms_select_us_per_pair_B{B}"] = 1e3 * (prof["nms_select"]["total_ms"] - prof["nms_select"]["count"] * ov) / 8 / (B // 2)
    kp = eng.kpts[:2, :4096].cpu().numpy()
    res["kpts_sha"] = hashlib.sha1(kp.tobytes()).hexdigest()[:16]
    # the stage entry point on the score map of the forward (no candidate epilogue)
    smap = torch.empty(2 * 1080 * 1920, device="cuda")
    eng.ctx.call("im_debug_read", b"sp_smap", smap.data_ptr(), smap.numel(), eng.stream_ptr())
    out = torch.empty_like(smap)
    for r in (4, 3):
        for _ in range(3):
            eng.ctx.call("im_nms", ptr(smap), ptr(out), 2, 1080, 1920, r, eng.stream_ptr())
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(reps):
            eng.ctx.call("im_nms", ptr(smap), ptr(out), 2, 1080, 1920, r, eng.stream_ptr())
        t1.record()
        torch.cuda.synchronize()
        res[f"im_nms_r{r}_us"] = 1e3 * t0.elapsed_time(t1) / reps
        res[f"map_r{r}_sha"] = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16]
    print(json.dumps(res))


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    for name, env in FORMS:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(reps)], env=dict(os.environ, **env), capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(f"{name}: " + (line[-1] if line else "FAILED " + r.stderr[-800:]), flush=True)


if __name__ == "__main__":
    child(int(sys.argv[2])) if "--child" in sys.argv else main()
