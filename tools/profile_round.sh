#!/bin/bash
# GPU box: the measurements DESIGN.md / profiles/ quote.  usage: tools/profile_round.sh <tag>
#   1. bench.py (full, with the CPU baseline leg)                 -> gpurun_out/prof_<tag>/bench.json
#   2. rocprofv3 --kernel-trace --stats of the same workload (config 2 only: the companion configs launch the same kernels on other planes) -> .../kernel_stats.csv (+ bench_under_rocprof.json)
#   3. separate --pmc passes (FETCH_SIZE, WRITE_SIZE)              -> .../pmc_hbm.json   (tools/pmc_summary.py)
tag=${1:-x}
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
out=gpurun_out/prof_$tag; mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err
tail -c 600 $out/bench.json
# the driver's command line (round-end BENCH): --steps 20 --warmup 5
python bench.py --steps 20 --warmup 5 > $out/bench_steps20_warmup5.json 2>> $out/bench.err
rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --chains 1 > $out/bench_under_rocprof.json 2> $out/rocprof.err
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
head -8 $out/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $out/pmc_$c -o p --output-format csv -- python bench.py --no-cpu-baseline --no-extra-configs --steps 10 --warmup 2 --chains 1 > /dev/null 2> $out/pmc_$c.err
done
python tools/pmc_summary.py $out/pmc_hbm.json $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
rm -rf $out/kt $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
