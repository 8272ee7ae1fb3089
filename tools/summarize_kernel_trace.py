#!/usr/bin/env python3
"""Summary of a rocprofv3 --kernel-trace CSV that separates a kernel's own duration from time-sharing with another queue.

`bench.py` keeps two pages in flight (detect / segment of page i+1 on a worker thread and its own HIP stream beside the diffusion steps
of page i), so in a trace of the default run a FLUX kernel's begin-to-end time includes the intervals in which kernels of the other
queue held part of the chip; rocprofv3's own `kernel_stats.csv` averages those in.  Per kernel name this prints calls, mean, median,
and the mean over the dispatches that did NOT overlap any dispatch of another queue — the figure that is comparable with the
in-context stamps of `bench.py` (which are taken with nothing else running).

usage: summarize_kernel_trace.py <kernel_trace.csv> [out.json] [top N]"""
import csv
import json
import sys
from bisect import bisect_left, bisect_right
from collections import defaultdict


def main():
    path = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    rows = []
    with open(path, newline="") as f:
        rd = csv.DictReader(f)
        for r in rd:
            rows.append((r["Kernel_Name"], int(r["Queue_Id"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    by_queue = defaultdict(list)
    for name, q, s, e in rows:
        by_queue[q].append((s, e))
    starts, ends = {}, {}
    for q, v in by_queue.items():
        v.sort()
        starts[q] = [s for s, _ in v]
        ends[q] = sorted(e for _, e in v)

    def overlapped(q, s, e):
        for q2 in by_queue:
            if q2 == q:
                continue
            # dispatches of q2 with start < e minus those with end <= s  > 0  <=> one of them intersects (s, e)
            if bisect_left(starts[q2], e) - bisect_right(ends[q2], s) > 0:
                return True
        return False

    agg = defaultdict(lambda: [[], []])
    for name, q, s, e in rows:
        a = agg[name]
        a[0].append(e - s)
        if not overlapped(q, s, e):
            a[1].append(e - s)
    total = sum(sum(a[0]) for a in agg.values())
    table = []
    for name, (al, alone) in sorted(agg.items(), key=lambda kv: -sum(kv[1][0]))[:top]:
        al_s = sorted(al)
        table.append({"kernel": name[:120], "calls": len(al), "share_of_gpu_time": sum(al) / total, "mean_us": sum(al) / len(al) / 1e3,
                      "median_us": al_s[len(al_s) // 2] / 1e3, "calls_not_sharing_the_chip": len(alone),
                      "mean_us_not_sharing": (sum(alone) / len(alone) / 1e3) if alone else None})
    rep = {"trace": path.split("/")[-1], "queues": {str(q): len(v) for q, v in by_queue.items()}, "kernels": table}
    if out:
        with open(out, "w") as f:
            json.dump(rep, f, indent=1)
    for t in table[:12]:
        print(f'{t["share_of_gpu_time"] * 100:5.1f}%  {t["calls"]:7d}  mean {t["mean_us"]:9.1f}  median {t["median_us"]:9.1f}  alone {t["mean_us_not_sharing"] or 0:9.1f} us ({t["calls_not_sharing_the_chip"]})  {t["kernel"][:70]}')


if __name__ == "__main__":
    main()
