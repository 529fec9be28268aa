#!/usr/bin/env python3
"""Instruction histogram of the MFMA-carrying basic blocks of a kernel in a hipcc -S listing.
Usage: python tools/isa_hist.py file.s <kernel-symbol-substring> [min_mfma]"""
import collections
import re
import sys


def main():
    path, sym = sys.argv[1], sys.argv[2]
    min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_][\w$.]*:", l) and sym in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur = [], ["entry", []]
    for l in lines[start:end]:
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append(cur)
            cur = [l.split(":")[0], []]
        elif l.startswith("\t") and not l.strip().startswith((".", ";")):
            cur[1].append(l.strip().split()[0])
    blocks.append(cur)
    for name, ins in blocks:
        c = collections.Counter(ins)
        mf = sum(v for k, v in c.items() if "mfma" in k)
        if mf < min_mfma:
            continue
        valu = sum(v for k, v in c.items() if k.startswith("v_") and "mfma" not in k)
        lds = sum(v for k, v in c.items() if k.startswith("ds_"))
        vmem = sum(v for k, v in c.items() if k.startswith(("global_", "buffer_")))
        print(f"{name}: {len(ins)} instructions, mfma {mf}, other VALU {valu}, LDS {lds}, VMEM {vmem}")
        print("   ", ", ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1]) if "mfma" not in k))


if __name__ == "__main__":
    main()
