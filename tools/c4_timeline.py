#!/usr/bin/env python
"""Timeline of config 4's shard (4 x 1 x 320 x 320, LADMM + CG + FFDNet-gray) from a rocprofv3 kernel trace of tools/bench_c4.py 4: the kernels of the
last two outer iterations with their durations and the gaps in front of them.  usage: c4_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_pnp_tail" in r["Kernel_Name"] or "k_bx_unpack_out" in r["Kernel_Name"]]
a, b = starts[-3], starts[-1]
sel = rows[a:b]
t0 = int(sel[0]["Start_Timestamp"])
busy, gaps, prev_end = 0, 0, int(rows[a - 1]["End_Timestamp"])
last = None
run = 0
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    gap = s - prev_end
    gaps += max(gap, 0)
    name = r["Kernel_Name"].replace("void ", "").replace("dpx::", "").split("(")[0][:50]
    if name == last and gap < 3000:
        run += 1
    else:
        if run:
            print(f"            ... x {run} more")
        run = 0
        print(f"{(s - t0) / 1e3:9.1f} us  gap {gap / 1e3:7.1f}  run {(e - s) / 1e3:7.1f}  {name}")
    last = name
    prev_end = max(prev_end, e)
span = int(rows[b]["Start_Timestamp"]) - t0
print(f"two outer iterations: span {span / 1e3:.1f} us, busy {busy / 1e3:.1f} us, gaps {gaps / 1e3:.1f} us")
