"""Static instruction census of one kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only), per basic block:
non-MFMA vector instructions, MFMAs, LDS and memory instructions, and the branch targets, to tell loop bodies from the
prologue / epilogue. fp32 MFMA and VALU never co-execute on MI355X, so the dynamic VALU count is what the MFMA pipe waits for.

    python tools/isa_census.py file.s conv3x3_wino_kernelILb1ELb0 [--blocks]
"""
import collections
import re
import sys


def main():
    path, sym = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(sym), l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    seg, cur, name = [], [], "entry"
    for l in lines[start:end]:
        if re.match(r"^\.LBB\d+_\d+:", l):
            seg.append((name, cur)); cur = []; name = l.split(":")[0]
        elif l.startswith("\t") and not l.strip().startswith((".", ";")):
            cur.append(l.strip())
            if l.strip().startswith(("s_cbranch", "s_branch")):      # what follows a branch is another block (the fall-through)
                seg.append((name, cur)); cur = []; name = name.split("+")[0] + "+"
    seg.append((name, cur))
    tot = collections.Counter()
    for n, c in seg:
        cc = collections.Counter(x.split()[0] for x in c)
        tot.update(cc)
        if "--blocks" in sys.argv:
            valu = sum(v for k, v in cc.items() if k.startswith("v_") and "mfma" not in k)
            br = [x for x in c if x.startswith(("s_cbranch", "s_branch"))]
            print(f"{n:10s} n={len(c):4d} valu={valu:4d} mfma={sum(v for k, v in cc.items() if 'mfma' in k):3d} "
                  f"ds={sum(v for k, v in cc.items() if k.startswith('ds_')):3d} "
                  f"vmem={sum(v for k, v in cc.items() if k.startswith(('buffer_', 'global_'))):3d} | "
                  + " ".join(b.split()[0][2:] + "->" + b.split()[-1] for b in br))
            print("           " + ", ".join(f"{k} {v}" for k, v in sorted(cc.items(), key=lambda kv: -kv[1])
                                            if k.startswith("v_") and "mfma" not in k)[:260])
    valu = sum(v for k, v in tot.items() if k.startswith("v_") and "mfma" not in k)
    print(f"{sym}: static instructions {sum(tot.values())}, non-MFMA VALU {valu}, MFMA {sum(v for k, v in tot.items() if 'mfma' in k)}")
    print("  " + ", ".join(f"{k} {v}" for k, v in sorted(tot.items(), key=lambda kv: -kv[1]) if k.startswith("v_") and "mfma" not in k)[:700])


if __name__ == "__main__":
    main()
