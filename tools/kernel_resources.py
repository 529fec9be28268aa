#!/usr/bin/env python3
"""Registers, spills and static LDS of every kernel in gfx950 assembly listings (hipcc -S --cuda-device-only), with the waves per SIMD
the register count allows (512 / VGPRs, at most 8). A latency- or HBM-bound kernel above 128 VGPRs runs at one or two waves per SIMD.

    python tools/kernel_resources.py file.s [file.s ...] [--min-vgpr N]
"""
import re
import sys


def main():
    thr = int(sys.argv[sys.argv.index("--min-vgpr") + 1]) if "--min-vgpr" in sys.argv else 0
    files = [a for a in sys.argv[1:] if a.endswith(".s")]
    for fn in files:
        t = open(fn).read()
        for blk in re.split(r"\n  - \.agpr_count:", t)[1:]:
            def g(k):
                m = re.search(r"\." + k + r":\s+(\S+)", blk)
                return m.group(1) if m else "0"
            v, a = int(g("vgpr_count")), int(re.match(r"\s*(\d+)", blk).group(1))
            name = re.sub(r"^_ZN2im\d+", "", g("name"))[:52]
            if v >= thr or int(g("vgpr_spill_count")):
                print(f"{fn.split('/')[-1]:16s} {name:52s} vgpr {v:3d} (agpr {a:3d}) spill {int(g('vgpr_spill_count')):4d} sgpr-spill {int(g('sgpr_spill_count')):3d} "
                      f"lds {int(g('group_segment_fixed_size')):6d}  waves/SIMD <= {min(8, 512 // max(v, 1))}")


if __name__ == "__main__":
    main()
