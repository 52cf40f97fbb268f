#!/usr/bin/env python3
"""The launch chain of the small levels from a rocprofv3 kernel trace of `bench.py --streams 1 --separate-calls`:
python tools/chain_timeline.py <kernel_trace.csv>
For the last track call: every dispatch in order with its duration and the gap to the previous dispatch's end, summarised per
(kernel, grid size): count, mean duration, mean gap before it."""
import csv
import re
import sys
from collections import defaultdict

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Grid_Size_Y", 1) or 1)) for r in rows)
segs, cur, cur_end = [], [], None
for e in ev:
    if cur and e[0] - cur_end > 60_000:
        segs.append(cur)
        cur = []
        cur_end = None
    cur.append(e)
    cur_end = e[1] if cur_end is None else max(cur_end, e[1])
if cur:
    segs.append(cur)
is_pose0 = lambda n: "eval_kernel<0, true" in n
idx = [i for i, sg in enumerate(segs) if any(is_pose0(e[2]) for e in sg)]
seg = segs[idx[-1]]
def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n.replace("dsm::", "")
agg = defaultdict(lambda: [0, 0, 0])
prev_end = None
t0 = seg[0][0]
seq = []
for s, e, n, gx, gy in seg:
    key = (short(n), gx, gy)
    a = agg[key]
    a[0] += 1
    a[1] += e - s
    if prev_end is not None:
        a[2] += s - prev_end
    seq.append((round((s - t0) / 1e3, 1), short(n)[:28], gx, gy, round((e - s) / 1e3, 1), round((s - prev_end) / 1e3, 1) if prev_end else 0))
    prev_end = e
print(f"track call: {len(seg)} dispatches, {(seg[-1][1] - t0) / 1e6:.3f} ms")
for (n, gx, gy), (c, tot, gap) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n[:40]:40s} grid {gx:7d} x {gy:4d}  n={c:4d}  mean {tot / c / 1e3:8.1f} us  gap before {gap / c / 1e3:6.1f} us  total {tot / 1e6:7.3f} ms")
print("first 60 dispatches (offset us, kernel, grid x, grid y, duration us, gap us):")
for q in seq[:60]:
    print("  ", q)
