#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
out=gpurun_out/c5tl; mkdir -p $out
rocprofv3 --kernel-trace -d $out/kt -o kt --output-format csv -- python tools/bench_configs.py c5 > $out/c5.log 2>&1
python tools/c5_timeline.py $(find $out/kt -name "*kernel_trace.csv" | head -1) > $out/timeline.txt; rm -rf $out/kt
cat $out/timeline.txt
