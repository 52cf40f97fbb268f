#!/usr/bin/env python3
"""Prices a range of an `hipcc -S` listing with the issue costs measured by tools/microbench/valu_issue_cost.hip
(profiles/r05_valu_issue_cost.json): cycles a wave64 instruction occupies its SIMD, by FORM --
  2.5  v_mul/add/sub/fmac/mov/and/or/xor/add_u32 ... in the 4-byte encoding with VGPR or inline-constant sources
  3.1  8-byte encodings of those and v_fma_f32 (VOP3), 32-bit literals
  4.65 anything with an SGPR source, every v_cmp / v_cndmask / v_cvt / v_fract / v_floor / v_frexp / shifts / integer mul, mad / med3 / DPP
  4.25 v_pk_*        8.5 v_rcp / v_rsq / v_sqrt ...
    python tools/isa_price.py t.s FIRST_LINE LAST_LINE [skip_label ...]     (blocks that start at a skip label are left out)"""
import collections
import re
import sys

FAST = ("v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fmac_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32",
        "v_sub_u32", "v_subrev_u32", "v_max_f32", "v_min_f32", "v_mov_b64")
SLOW = ("v_cmp", "v_cndmask", "v_cvt", "v_fract", "v_floor", "v_frexp", "v_lshl", "v_lshr", "v_ashr", "v_mul_lo", "v_mul_hi", "v_mad_u32", "v_mad_i32",
        "v_add_lshl", "v_lshl_add", "v_med3", "v_min3", "v_max3", "v_min_i32", "v_max_i32", "v_min_u32", "v_readlane", "v_readfirstlane", "v_writelane",
        "v_bfe", "v_div_scale", "v_div_fmas", "v_div_fixup", "v_ldexp", "v_trunc", "v_rndne")
TRANS = ("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")


def price(op, args):
    has_s = re.search(r"(?<![a-z_\[])s\d+|s\[\d+:\d+\]|vcc|exec", args) is not None
    lit = re.search(r"0x[0-9a-f]{5,}", args) is not None
    if op.startswith(TRANS):
        return 8.5, "transcendental"
    if op.startswith("v_pk_"):
        return 4.25, "packed"
    if "_dpp" in op or "quad_perm" in args or "row_" in args:
        return 4.65, "dpp"
    if op.startswith(SLOW):
        return 4.65, "cmp / select / convert / shift / integer"
    if has_s:
        return 4.65, "SGPR source"
    if op.startswith(FAST) and op.endswith("_e32") and not lit:
        return 2.5, "fast (e32, VGPR or inline sources)"
    if op.startswith(FAST) or op.startswith("v_fma_f32"):
        return 3.1, "8-byte encoding / literal"
    return 4.65, "other: " + op


def main():
    path, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    skip = set(sys.argv[4:])
    lines = open(path).read().splitlines()[a - 1:b]
    tot = collections.Counter()
    cnt = collections.Counter()
    skipping = False
    for l in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            skipping = m.group(1) in skip
            continue
        if skipping or not l.startswith("\t"):
            continue
        t = l.strip().split(None, 1)
        op, args = t[0], (t[1] if len(t) > 1 else "")
        if not op.startswith("v_"):
            continue
        c, why = price(op, args.split(";")[0])
        tot[why] += c
        cnt[why] += 1
    n = sum(cnt.values())
    print(f"{n} VALU instructions, {sum(tot.values()):.0f} SIMD cycles")
    for k, v in tot.most_common():
        print(f"  {k:45s} {cnt[k]:4d} instr  {v:7.0f} cycles")


if __name__ == "__main__":
    main()
