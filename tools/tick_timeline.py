#!/usr/bin/env python3
"""Timeline of a `bench.py --stream 1 --stream-engine 1` run from a rocprofv3 kernel trace:
   python tools/tick_timeline.py <kernel_trace.csv> [n_last_advances]
Advances are separated by the host's read-back.  For the last advances: duration, time with some kernel running, and per
kernel the dispatches, summed / union time and average duration; then the host gap between consecutive advances."""
import csv
import re
import sys
from collections import defaultdict

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
segs, cur, cur_end = [], [], None
for e in ev:
    if cur and e[0] - cur_end > 20_000:
        segs.append(cur)
        cur = []
        cur_end = None
    cur.append(e)
    cur_end = e[1] if cur_end is None else max(cur_end, e[1])
if cur:
    segs.append(cur)
segs = [sg for sg in segs if sum("tick_eval_kernel" in e[2] for e in sg) >= 4]
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


def short(n):
    return re.sub(r"\(.*$", "", re.sub(r"^void ", "", n)).replace("dsm::", "")[:40]


prev_end = None
for sg in segs[-n_last:]:
    t0, t1 = sg[0][0], max(e[1] for e in sg)
    agg = defaultdict(lambda: [0, 0, []])
    for s, e, n in sg:
        a = agg[short(n)]
        a[0] += 1
        a[1] += e - s
        a[2].append((s, e))
    gap = f", host gap before it {(t0 - prev_end) / 1e3:.0f} us" if prev_end else ""
    prev_end = t1
    print(f"advance: {len(sg)} dispatches, {(t1 - t0) / 1e6:.3f} ms; some kernel running {union([(s, e) for s, e, _ in sg]) / 1e6:.3f} ms{gap}")
    for k, (c, tot, iv) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"    {k:40s} n={c:4d} sum={tot / 1e6:7.3f} ms union={union(iv) / 1e6:7.3f} ms avg={tot / c / 1e3:7.1f} us")
