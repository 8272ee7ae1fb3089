#!/usr/bin/env python3
"""How much of a rocprofv3 --kernel-trace is concurrent execution?  Prints, per queue, dispatch count and busy time, then the union of
all busy intervals against their sum (sum / union = average number of kernels in flight while anything runs) and the idle share.
usage: trace_overlap.py <kernel_trace.csv> [t0_fraction t1_fraction]   (fractions of the trace's span to look at, default 0.5 1.0)"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
lo, hi = rows[0][0], max(r[1] for r in rows)
f0, f1 = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.5, 1.0)
a, b = lo + (hi - lo) * f0, lo + (hi - lo) * f1
rows = [r for r in rows if r[0] >= a and r[1] <= b]
byq = defaultdict(lambda: [0, 0])
for s, e, q, _ in rows:
    byq[q][0] += 1
    byq[q][1] += e - s
for q, (n, t) in sorted(byq.items()):
    print(f"queue {q}: {n} dispatches, busy {t / 1e6:.2f} ms")
total = sum(e - s for s, e, _, _ in rows)
union, cur_s, cur_e = 0, None, None
for s, e, _, _ in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
span = rows[-1][1] - rows[0][0]
print(f"span {span / 1e6:.2f} ms, union of busy intervals {union / 1e6:.2f} ms ({100 * union / span:.1f} % of the span), sum of durations {total / 1e6:.2f} ms -> {total / union:.2f} kernels in flight on average")
