#!/usr/bin/env python
"""Timeline of the timed region of `bench.py --steps K` from a rocprofv3 kernel trace: span, busy time and the largest gaps of the
last 2K + 2 dpx kernels (seed rhs, seed row transform, K x (column kernel, row kernel)).  usage: trace_gaps.py <kernel_trace.csv> <K>"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
K = int(sys.argv[2])
idx = [i for i, r in enumerate(rows) if "k_iter_rows" in r["Kernel_Name"]]
last = idx[-1]
first = last - (2 * K + 2) + 1
sel = rows[first:last + 1]
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
print(f"{len(sel)} kernels, span {(t1 - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, gaps {(t1 - t0 - busy) / 1e3:.1f} us")
for a, b in zip(sel[:-1], sel[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g > 4000:
        print(f"  gap {g / 1e3:.1f} us between {a['Kernel_Name'][:40]} and {b['Kernel_Name'][:40]}")
for r in sel[:3] + sel[-3:]:
    print(f"  {r['Kernel_Name'][10:50]:42s} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f} us")
