#!/usr/bin/env python
"""Reads a rocprofv3 kernel trace (csv) of tools/plane_chain_probe.py and reports, for the last 200 launches of the two iteration kernels, how
their executions overlap: per queue the kernels it ran, and the fraction of the busy time during which 1, 2, 3 ... kernels were running."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_cols_p2" in r["Kernel_Name"] or "k_iter_rows" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -240:]
qs = {}
for r in rows:
    qs.setdefault(r.get("Queue_Id", "?"), []).append(r)
print({q: len(v) for q, v in qs.items()})
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
depth, last, hist = 0, ev[0][0], {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - last)
    depth += d; last = t
tot = sum(hist.values())
print("span %.1f us for %d launches; time with n kernels running:" % (tot / 1e3, len(rows)), {k: round(v / tot, 3) for k, v in sorted(hist.items())})
for r in rows[-12:]:
    print(r.get("Queue_Id"), (int(r["Start_Timestamp"]) - int(rows[-12]["Start_Timestamp"])) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:40], r.get("Grid_Size"), r.get("Workgroup_Size"))
