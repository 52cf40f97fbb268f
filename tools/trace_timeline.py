#!/usr/bin/env python3
"""Timeline of one bench step from a rocprofv3 kernel trace:  python tools/trace_timeline.py <kernel_trace.csv> [step_ms_hint]
Prints, for the LAST track+scale step of the run, per kernel symbol: dispatches, summed and union time, and a compact
sequence of the level-0 dispatches (start offset, duration)."""
import csv
import re
import sys
from collections import defaultdict

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# calls are separated by host gaps (> 60 us without any kernel running): split, keep the last track call (the last segment
# that contains level-0 POSE evaluations) and the scale call after it
segs, cur, cur_end = [], [], None
for e in ev:
    if cur and e[0] - cur_end > 60_000:
        segs.append(cur)
        cur = []
    cur.append(e)
    cur_end = e[1] if cur_end is None or not cur[:-1] else max(cur_end, e[1])
if cur:
    segs.append(cur)
is_pose0 = lambda n: "eval_kernel<0, true" in n
idx = [i for i, sg in enumerate(segs) if any(is_pose0(e[2]) for e in sg)]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # 0: the track call, 1: the call after it (scale optimisation)
seg = segs[min(len(segs) - 1, idx[-1] + which)]
t0 = seg[0][0]
def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n.replace("dsm::", "")
agg = defaultdict(lambda: [0, 0, []])
for s, e, n in seg:
    a = agg[short(n)]
    a[0] += 1
    a[1] += e - s
    a[2].append((s, e))
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
print(f"segment: {len(seg)} dispatches over {(seg[-1][1] - t0) / 1e6:.3f} ms; all-kernel union {union([(s, e) for s, e, _ in seg]) / 1e6:.3f} ms")
for n, (c, tot, iv) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n[:70]:70s} n={c:5d} sum={tot / 1e6:8.3f} ms union={union(iv) / 1e6:8.3f} ms avg={tot / c / 1e3:8.1f} us")
big = [(s, e, n) for s, e, n in seg if is_pose0(n)]
print("level-0 eval dispatches (offset ms, duration us):", [(round((s - t0) / 1e6, 3), round((e - s) / 1e3)) for s, e, n in big][:40])
