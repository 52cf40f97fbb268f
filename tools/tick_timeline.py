#!/usr/bin/env python3
"""Timeline of a `bench.py --stream 1 --stream-engine 1` run from a rocprofv3 kernel trace:
   python tools/tick_timeline.py <kernel_trace.csv> [n_last_advances [gantt_us]]
Advances are separated by the host's read-back.  For the last advances: duration, time with some kernel running, and per
kernel the dispatches, summed / union time and average duration; then the host gap between consecutive advances."""
import csv
import re
import sys
from collections import defaultdict

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
queue_of = {(int(r["Start_Timestamp"]), int(r["End_Timestamp"])): r.get("Queue_Id", "?") for r in rows}
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


# Pipelined advances leave no host gap between them: besides the gap-separated segments, the same table over a steady-state WINDOW --
# the middle fifth of the run's tick_eval dispatches
te = [e for e in ev if "tick_eval_kernel<0" in e[2]]
if len(te) > 50:
    w0, w1 = te[int(0.4 * len(te))][0], te[int(0.6 * len(te))][1]
    win = [e for e in ev if e[0] >= w0 and e[1] <= w1]
    agg = defaultdict(lambda: [0, 0, []])
    for s_, e_, n_ in win:
        a = agg[short(n_)]
        a[0] += 1
        a[1] += e_ - s_
        a[2].append((s_, e_))
    n_ticks = sum(1 for e in win if "tick_lm_kernel<1>" in e[2]) or 1
    print(f"steady-state window: {(w1 - w0) / 1e6:.3f} ms, {len(win)} dispatches, some kernel running {union([(a, b) for a, b, _ in win]) / 1e6:.3f} ms; "
          f"{n_ticks} ticks -> {(w1 - w0) / 1e3 / n_ticks:.1f} us per tick")
    for k, (c, tot, iv) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"    {k:40s} n={c:4d} sum={tot / 1e6:7.3f} ms union={union(iv) / 1e6:7.3f} ms avg={tot / c / 1e3:7.1f} us")
if len(sys.argv) > 3 and len(te) > 50:  # every dispatch of a short window in the middle of the run, with its queue: who waits for whom
    g0 = te[len(te) // 2][0]
    g1 = g0 + int(float(sys.argv[3]) * 1e3)
    qs = sorted({queue_of[(a, b)] for a, b, _ in ev if g0 <= a < g1})
    print(f"gantt: {float(sys.argv[3]):.0f} us from the middle of the run; columns: queue, start us, end us, duration us, kernel")
    for a, b, n_ in ev:
        if g0 <= a < g1:
            q = qs.index(queue_of[(a, b)])
            print(f"    q{q} {'  ' * q}{(a - g0) / 1e3:8.1f} {(b - g0) / 1e3:8.1f} {(b - a) / 1e3:7.1f}  {short(n_)}")
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
