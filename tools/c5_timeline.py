#!/usr/bin/env python
"""Timeline of config 5's last training step from a rocprofv3 kernel trace (tools/c5_timeline.sh): every kernel of the step with its
duration and the gap in front of it.  usage: c5_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# a step starts at the first forward column kernel after a backward finishing kernel: find the last k_solve_rhs_bwd (end of a backward pass)
ends = [i for i, r in enumerate(rows) if "k_solve_rhs_bwd" in r["Kernel_Name"]]
if len(ends) < 2:
    sys.exit("no two training steps in the trace")
a, b = ends[-2], ends[-1]
sel = rows[a + 1:b + 3]
t0 = int(sel[0]["Start_Timestamp"])
busy = 0
prev_end = int(rows[a]["End_Timestamp"])
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    name = r["Kernel_Name"].replace("void ", "").replace("dpx::", "")[:44]
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f}  run {(e - s) / 1e3:7.1f}  {name}")
    prev_end = max(prev_end, e)
span = int(sel[-1]["End_Timestamp"]) - t0
print(f"span {span / 1e3:.1f} us, busy {busy / 1e3:.1f} us")
