#!/usr/bin/env python3
"""Per-pass timeline of a `bench.py --stream 1` run from a rocprofv3 kernel trace:
   python tools/stream_timeline.py <kernel_trace.csv> [n_last_passes]
Passes are separated by the host's read-back (a gap without any kernel).  For each of the last passes: duration, the time at
least one kernel ran (union), and per kernel class the summed / union time and the number of dispatches; then how much of the
pass had a LARGE kernel (level 0 / 1 evaluation) running, a small one only, or nothing."""
import csv
import re
import sys
from collections import defaultdict

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Grid_Size_Y", 1) or 1)) for r in rows)
segs, cur, cur_end = [], [], None
for e in ev:
    if cur and e[0] - cur_end > 25_000:
        segs.append(cur)
        cur = []
        cur_end = None
    cur.append(e)
    cur_end = e[1] if cur_end is None else max(cur_end, e[1])
if cur:
    segs.append(cur)
segs = [sg for sg in segs if any("eval_kernel<0, true" in e[2] for e in sg)]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 4


def union(iv):
    iv = sorted(iv)
    busy, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return busy + (ce - cs if cs is not None else 0)


def klass(n, gy):
    m = re.search(r"eval_kernel<(\d), (true|false), (true|false)(?:, (\d))?", n)
    if m:
        mode = "pose" if m.group(1) == "0" else "scale"
        return f"eval {mode} {'L0' if m.group(2) == 'true' else 'L>=1'}{' fused' if m.group(3) == 'true' else ''}{' ro' + m.group(4) if m.group(4) and m.group(4) != '0' else ''}"
    if "lm_kernel" in n:
        return "lm_kernel"
    return re.sub(r"\(.*$", "", re.sub(r"^void ", "", n)).replace("dsm::", "")[:40]


for sg in segs[-n_last:]:
    t0, t1 = sg[0][0], max(e[1] for e in sg)
    agg = defaultdict(lambda: [0, 0, []])
    for s, e, n, gx, gy in sg:
        a = agg[klass(n, gy)]
        a[0] += 1
        a[1] += e - s
        a[2].append((s, e))
    big = [(s, e) for s, e, n, gx, gy in sg if "eval_kernel<0, true" in n or (e - s) > 100_000]
    allk = [(s, e) for s, e, *_ in sg]
    print(f"pass: {len(sg)} dispatches, {(t1 - t0) / 1e6:.3f} ms; some kernel running {union(allk) / 1e6:.3f} ms; a level-0 (or > 100 us) kernel running {union(big) / 1e6:.3f} ms")
    for k, (c, tot, iv) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"    {k:32s} n={c:4d} sum={tot / 1e6:7.3f} ms union={union(iv) / 1e6:7.3f} ms avg={tot / c / 1e3:7.1f} us")
