#!/bin/bash
# GPU box: config 4's shard (4 x 1 x 320 x 320) under the A/B switches of its loop + a kernel timeline of the default.  -> gpurun_out/c4b/
mkdir -p gpurun_out/c4b; cd /root/repo
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cg or csmri or c4 or ladmm" > gpurun_out/c4b/tests.log 2>&1; tail -3 gpurun_out/c4b/tests.log
for i in 1 2; do
echo "--- default (tag polling, one call per iteration, folded tail)"; python tools/bench_c4.py 4 32 2>&1 | tail -2
echo "--- folded, head not issued early"; DPX_PNP_CG_NO_FOLD=2 python tools/bench_c4.py 4 2>&1 | tail -1
echo "--- head / tail not folded"; DPX_PNP_CG_NO_FOLD=1 python tools/bench_c4.py 4 2>&1 | tail -1
echo "--- staged loop"; DPX_SPLIT_CG_STAGED=1 python tools/bench_c4.py 4 2>&1 | tail -1
echo "--- event wait, one call"; DPX_CG_EVENT_WAIT=1 python tools/bench_c4.py 4 2>&1 | tail -1
echo "--- event wait, staged (round 4)"; DPX_CG_EVENT_WAIT=1 DPX_SPLIT_CG_STAGED=1 python tools/bench_c4.py 4 2>&1 | tail -1
done
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c4b/tr -- python $R/tools/bench_c4.py 4 > /dev/null 2>&1
cd $R; f=$(ls gpurun_out/c4b/tr/*/*kernel_trace.csv | head -1); python tools/c4_timeline.py $f > gpurun_out/c4b/timeline.txt; rm -rf gpurun_out/c4b/tr; tail -40 gpurun_out/c4b/timeline.txt
