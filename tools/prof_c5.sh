#!/bin/bash
# GPU box: where config 5's training step spends its time (kernel stats + host profile)
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
out=gpurun_out/c5; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python tools/bench_configs.py c5 > $out/c5.log 2>&1
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv; rm -rf $out/kt
head -25 $out/kernel_stats.csv; tail -2 $out/c5.log
DPX_C5_PROFILE=1 python tools/bench_configs.py c5 2>&1 | tail -45
