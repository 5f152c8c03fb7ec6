#!/usr/bin/env python
"""Timeline of the `bench.py --steps K` timed regions from a rocprofv3 kernel trace, chains included: every run of the two-kernel
iteration (seed launches, then column / row launches) with its kernel count and span; for the LAST run with K iterations, when its first
and last kernels start / end relative to its first kernel.  usage: region_timeline.py <kernel_trace.csv> <K>"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
K = int(sys.argv[2])
it = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_seed_rows", "k_cols_p2<1024, 64, 8, 2", "k_iter_rows"))]
runs, cur = [], []
for r in it:
    if "k_seed_rows" in r["Kernel_Name"] and cur and "k_seed_rows" not in cur[-1]["Kernel_Name"]:
        runs.append(cur)
        cur = []
    cur.append(r)
if cur:
    runs.append(cur)
pick = None
for run in runs:
    ns = sum("k_seed_rows" in r["Kernel_Name"] for r in run)
    n_it = (len(run) - ns) // (2 * max(ns, 1))
    span = (max(int(r["End_Timestamp"]) for r in run) - int(run[0]["Start_Timestamp"])) / 1e3
    print(f"run: {ns} chain(s), {n_it:3d} iterations, span {span:9.1f} us = {span / max(n_it, 1):7.2f} us per iteration")
    if n_it == K:
        pick = run
if pick:
    t0 = int(pick[0]["Start_Timestamp"])
    key = "Queue_Id" if "Queue_Id" in pick[0] else None
    if key:
        ends = {}
        for r in pick:
            ends[r[key]] = max(ends.get(r[key], 0), int(r["End_Timestamp"]))
        print("  last kernel of each queue ends at", {q: round((e - t0) / 1e3, 1) for q, e in ends.items()}, "us")
    for r in pick[:12] + pick[-8:]:
        print(f"  {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} .. {(int(r['End_Timestamp']) - t0) / 1e3:9.1f} us  q={r.get(key, '?') if key else '?'}  {r['Kernel_Name'].split('(')[0][-44:]}")
