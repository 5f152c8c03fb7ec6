#!/usr/bin/env python
"""Timeline of the LAST `bench.py --steps K` timed region from a rocprofv3 kernel trace, chains included: when each of the region's first
and last kernels starts / ends relative to the region's first kernel, and the busy time per stream.  usage: region_timeline.py <kernel_trace.csv> <K>"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
K = int(sys.argv[2])
dpx = [r for r in rows if "dpx::" in r["Kernel_Name"]]
# the last region: walk back from the last k_iter_rows launch to the seed launches in front of it
seeds = [i for i, r in enumerate(dpx) if "k_seed_rows" in r["Kernel_Name"]]
last_rows = max(i for i, r in enumerate(dpx) if "k_iter_rows" in r["Kernel_Name"])
first = max(i for i in seeds if i < last_rows)
while first - 1 in seeds:
    first -= 1
sel = dpx[first:last_rows + 1]
t0 = int(sel[0]["Start_Timestamp"])
print(f"{len(sel)} kernels, span {(int(sel[-1]['End_Timestamp']) - t0) / 1e3:.1f} us = {(int(sel[-1]['End_Timestamp']) - t0) / 1e3 / K:.2f} us per step")
key = "Queue_Id" if "Queue_Id" in sel[0] else ("Stream_Id" if "Stream_Id" in sel[0] else None)
for r in sel[:10] + sel[-6:]:
    print(f"  {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} .. {(int(r['End_Timestamp']) - t0) / 1e3:9.1f} us  q={r.get(key, '?') if key else '?'}  {r['Kernel_Name'].split('(')[0][-40:]}")
