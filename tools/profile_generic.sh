#!/bin/bash
# GPU box: rocprofv3 kernel statistics + HBM counters of the ADMM iteration on a plane off the register-radix path (8x3x1000x1000)
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
out=gpurun_out/prof_generic; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python tools/bench_shapes.py 8x3x1000x1000 > $out/run.log 2>&1
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv; rm -rf $out/kt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $out/pmc_$c -o p --output-format csv -- python tools/bench_shapes.py 8x3x1000x1000 > /dev/null 2> $out/pmc_$c.err
done
python tools/pmc_summary.py $out/pmc_hbm.json $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE > $out/pmc_summary.txt
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
head -8 $out/kernel_stats.csv | cut -c1-200; cat $out/pmc_summary.txt
