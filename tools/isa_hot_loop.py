#!/usr/bin/env python3
"""Finds the two-points-per-trip evaluation loop (the block holding >= 72 v_fmac_f32 and the blocks from its loop header on) of a
kernel in an `hipcc -S` listing and prices its common path with tools/isa_price.py's table.
    python tools/isa_hot_loop.py t.s <mangled kernel prefix> [lo hi]     (blocks holding lo <= v_fmac_f32 < hi; default 72 .. 999:
                                                                           the two-point loop; 40 72: the one-point loops)"""
import re
import subprocess
import sys
import os


def main():
    f, kn = sys.argv[1], sys.argv[2]
    lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (72, 999)
    lines = open(f).read().splitlines()
    k = next(i for i, l in enumerate(lines) if l.startswith(kn))
    e = next(i for i in range(k, len(lines)) if lines[i].startswith(".Lfunc_end"))
    labs = [(i, l.split(":")[0]) for i, l in enumerate(lines[k:e], k) if re.match(r"^\.LBB\d+_\d+:", l)] + [(e, "")]
    found = False
    for n, (i, lab) in enumerate(labs[:-1]):
        j = labs[n + 1][0]
        if lo <= sum(1 for x in lines[i:j] if "v_fmac_f32" in x) < hi:
            br = [l for l in lines[i:j] if re.match(r"^\s+s_branch", l)]
            tgt = br[-1].split()[-1] if br else lab
            hdr = next(a for a, b in labs if b == tgt)
            print(f"{kn}: loop header {tgt} line {hdr + 1} .. {j}")
            sys.stdout.flush()
            subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "isa_price.py"), f, str(hdr + 1), str(j)])
            found = True
    if not found:
        print("no such block found")


if __name__ == "__main__":
    main()
