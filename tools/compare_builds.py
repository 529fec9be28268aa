#!/usr/bin/env python3
"""Are two builds of the library bit-identical on the benchmark pair? Runs SuperPoint + LightGlue (1080 x 1920 stereo pair and the
translated pair, 4096 keypoints) with each library named on the command line in its own process and prints a SHA-1 per output
(keypoints, scores, descriptors, matches0, matching_scores0): equal digests = equal bits. Used after instruction-level rewrites that
must not change any result (same operations in the same order).

    python tools/compare_builds.py build_abl/<old>/libicematch.so icepy4d_amd/csrc/libicematch.so
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from icepy4d_amd import synthetic
    from icepy4d_amd.engine import Engine
    eng = Engine(0)
    eng.load_state_dict("superpoint", synthetic.superpoint_state_dict(0))
    eng.load_state_dict("lightglue", synthetic.lightglue_state_dict(0, "passthrough"))
    out = {}
    for name, (a, b) in (("stereo", synthetic.stereo_pair(0, 1080, 1920)), ("translated", synthetic.translated_pair(0, 1080, 1920, 40, 8))):
        eng.reserve(1080, 1920, 2, 4096)
        eng.superpoint(torch.from_numpy(np.stack([a, b])).cuda(), max_kpts=4096)
        eng.lightglue((1920.0, 1080.0), (1920.0, 1080.0))
        k0, d0, s0 = eng.features_to_host(0)
        k1, d1, s1 = eng.features_to_host(1)
        m = eng.matches_to_host(len(k0), len(k1))
        for key, arr in (("kpts0", k0), ("kpts1", k1), ("scores0", s0), ("desc0", d0), ("desc1", d1), ("matches0", m["matches0"]),
                         ("mscores0", m["matching_scores0"])):
            out[f"{name}.{key}"] = hashlib.sha1(np.ascontiguousarray(arr).tobytes()).hexdigest()[:12]
        out[f"{name}.n_matches"] = int((np.asarray(m["matches0"]) >= 0).sum())
    print(json.dumps(out))


def main():
    rows = []
    for lib in sys.argv[1:]:
        env = dict(os.environ, ICEMATCH_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(lib, "FAILED", r.stderr[-800:]); sys.exit(1)
        rows.append(json.loads(line[-1]))
    same = True
    for k in rows[0]:
        vals = [r[k] for r in rows]
        flag = "" if len(set(map(str, vals))) == 1 else "   <-- DIFFERENT"
        same &= not flag
        print(f"{k:24s}" + "  ".join(str(v) for v in vals) + flag)
    print("bit-identical" if same else "NOT identical")
    sys.exit(0 if same else 2)


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
