#!/bin/bash
# GPU box: kernel-time summary of one tools/bench_configs.py configuration: tools/prof_cfg.sh c4
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
cfg=${1:-c4}; out=gpurun_out/$cfg; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python tools/bench_configs.py $cfg > $out/run.log 2>&1
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv; rm -rf $out/kt
python - <<PY
import csv
rows=list(csv.DictReader(open("$out/kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in rows[:14]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} total {float(r["TotalDurationNs"])/1e6:8.2f} ms avg {float(r["AverageNs"])/1e3:8.1f} us {r["Percentage"]:>6s}%')
PY
tail -2 $out/run.log
