#!/bin/bash
# GPU box: config 5's per-kernel times for a setting of the backward knobs (env passes through)
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
out=gpurun_out/c5q_$1; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python tools/bench_configs.py c5 > $out/c5.log 2>&1
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv; rm -rf $out/kt
grep config5 $out/c5.log
head -9 $out/kernel_stats.csv | cut -d, -f1-4 | sed 's/(.*)//' | cut -c1-120
