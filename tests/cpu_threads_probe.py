"""TEST INFRASTRUCTURE (imports oracle/): how many host threads the CPU baseline of bench.py should use on the GPU box
(SuperPoint of one 1080p image with 16 / 32 / 64 torch threads). Run from the repo root: python tests/cpu_threads_probe.py"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from icepy4d_amd import synthetic
from oracle import ref_cpu
sp = synthetic.superpoint_state_dict(0); lg = synthetic.lightglue_state_dict(0, "passthrough")
a, b = synthetic.stereo_pair(0, 1080, 1920)
for nt in (16, 32, 64):
    torch.set_num_threads(nt)
    s, t = synthetic.stereo_pair(0, 120, 160); ref_cpu.match_images_lightglue(s, t, sp, lg, max_keypoints=64)
    t0 = time.perf_counter()
    with torch.inference_mode():
        f0 = ref_cpu.superpoint_lg(ref_cpu.frame_to_tensor(a), sp, 4096)
    t1 = time.perf_counter()
    print(nt, "threads: superpoint one image", round(t1 - t0, 2), "s", flush=True)
    if nt == 32:
        with torch.inference_mode():
            f1 = ref_cpu.superpoint_lg(ref_cpu.frame_to_tensor(b), sp, 4096)
            t2 = time.perf_counter()
            ref_cpu.lightglue(f0, f1, lg)
        print(nt, "threads: lightglue", round(time.perf_counter() - t2, 2), "s", flush=True)
