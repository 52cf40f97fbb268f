#!/usr/bin/env python3
"""Lists the loops of one kernel in an `hipcc -S` listing with their instruction mix.
    hipcc --offload-arch=gfx950 ... -S --cuda-device-only -o t.s tracker_kernels.hip
    python tools/isa_loops.py t.s '<mangled kernel name prefix>' [min_instructions]
A loop = a backward branch to a label; the body = the lines between the label and the branch (inner blocks included)."""
import collections
import re
import sys


def main():
    path, name = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(name) and l.split(":")[0].startswith(name))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end + 1]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    for i, l in enumerate(body):
        m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"^\s+s_branch\s+(\.LBB\d+_\d+)", l)
        if not m or m.group(1) not in labels or labels[m.group(1)] >= i:
            continue
        j = labels[m.group(1)]
        ins = [x.split()[0] for x in body[j:i + 1] if x.startswith("\t") and not x.strip().startswith((";", "."))]
        if len(ins) < min_n:
            continue
        cat = collections.Counter()
        for op in ins:
            if op.startswith("v_pk_"):
                cat["valu_packed"] += 1
            elif op.startswith(("v_fma", "v_fmac", "v_mac")):
                cat["valu_fma"] += 1
            elif op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
                cat["valu_transcendental"] += 1
            elif op.startswith("v_cmp"):
                cat["valu_cmp"] += 1
            elif op.startswith(("v_cndmask", "v_mov", "v_and", "v_or", "v_xor", "v_lshl", "v_lshr", "v_ashr", "v_bfe", "v_readlane", "v_readfirstlane", "v_writelane", "v_accvgpr")):
                cat["valu_move_logic"] += 1
            elif op.startswith("v_"):
                cat["valu_other_arith"] += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                cat["vmem"] += 1
            elif op.startswith("ds_"):
                cat["lds"] += 1
            elif op.startswith("s_waitcnt"):
                cat["s_waitcnt"] += 1
            elif op.startswith("s_"):
                cat["salu"] += 1
            else:
                cat["other"] += 1
        valu = sum(v for k, v in cat.items() if k.startswith("valu"))
        print(f"loop {m.group(1)} lines {start + j + 1}-{start + i + 1}: {len(ins)} instructions, {valu} VALU  {dict(sorted(cat.items()))}")
        ops = collections.Counter(op for op in ins if op.startswith("v_"))
        print("   ", ", ".join(f"{k} {v}" for k, v in ops.most_common(40)))


if __name__ == "__main__":
    main()
