#!/bin/bash
# GPU box: the fixed parts of a short run after a change of the emitting / seeding passes.  usage: tools/emit_check.sh <tag> [pytest -k expr]
tag=${1:-x}; kexpr=${2:-"x_alone or fresh or nodual or band or full_c2 or config2"}
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "$kexpr" > $out/tests.log 2>&1; tail -3 $out/tests.log
for i in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > $out/bench20_$i.json 2>> $out/bench.err
  python - <<PY
import json
d = json.loads(open("$out/bench20_$i.json").read().strip().splitlines()[-1])
print("20/5:", d["value"], d["ms_per_step"], "steady", d.get("steady_state", {}).get("value"), "cold", d.get("cold_solve", {}))
PY
done
rocprofv3 --kernel-trace -d $out/kt -o kt --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > $out/bench_rocprof.json 2> $out/rocprof.err
f=$(find $out/kt -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f 20 | tee $out/gaps.log
python - <<PY | tee $out/emit_durations.log
import csv
rows = sorted(csv.DictReader(open("$f")), key=lambda r: int(r["Start_Timestamp"]))
import collections
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "dpx::" in n:
        d[n.split("(")[0][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in d.items():
    v2 = sorted(v)
    print(f"{n:72s} n={len(v):4d} med={v2[len(v2)//2]:7.1f} min={v2[0]:7.1f} max={v2[-1]:7.1f}  top5={[round(x,1) for x in v2[-5:]]}")
PY
rm -rf $out/kt
