#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV output (one directory per pass) into profiles/<name>.json:
per kernel the average counter values per launch.  HBM traffic per launch follows MI355X_MICROARCH.md:
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads
(doubled here, flagged as corrected); WRITE_SIZE is used as is (checked against known store volumes)."""
import collections, csv, glob, json, sys
out_path, dirs = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + "/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("dpx::", "")
            if not k.startswith("k_"):
                continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for k, v in acc.items():
    c = {n: sum(x) / len(x) for n, x in v.items()}
    e = {"launches_sampled": max(len(x) for x in v.values()), "counters": c}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["hbm_read_bytes_corrected"] = c["FETCH_SIZE"] * 1024 * 2
        e["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
        e["hbm_traffic_bytes"] = e["hbm_read_bytes_corrected"] + e["hbm_write_bytes"]
    res[k] = e
json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
for k, e in sorted(res.items()):
    if "hbm_traffic_bytes" in e:
        print(f"{k[:40]:40s} read {e['hbm_read_bytes_corrected']/1e6:8.1f} MB  write {e['hbm_write_bytes']/1e6:8.1f} MB")
