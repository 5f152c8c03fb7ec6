#!/bin/bash
# GPU box: the measurements DESIGN.md / profiles/ quote for round 4.  usage: tools/profile_round4.sh <tag>
tag=${1:-r4}
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
out=gpurun_out/prof_$tag; mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err
python bench.py --steps 20 --warmup 5 > $out/bench_steps20_warmup5.json 2>> $out/bench.err
# kernel trace of the driver's command (config 2 only, one chain: every launch has the GPU to itself)
rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-extra-configs --chains 1 > $out/bench_under_rocprof.json 2> $out/rocprof.err
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv; rm -rf $out/kt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $out/pmc_$c -o p --output-format csv -- python bench.py --pmc-child --no-pmc --no-cpu-baseline --no-extra-configs --steps 10 --warmup 2 --chains 1 > /dev/null 2> $out/pmc_$c.err
done
python tools/pmc_summary.py $out/pmc_hbm.json $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE > $out/pmc_summary.txt
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
# config 4 shard and config 5: kernel traces of the same library
rocprofv3 --kernel-trace --stats -d $out/kt4 -o kt --output-format csv -- python tools/bench_c4.py 4 > $out/c4shard.log 2>&1
cp $(find $out/kt4 -name "*kernel_stats.csv" | head -1) $out/c4shard_kernel_stats.csv; rm -rf $out/kt4
rocprofv3 --kernel-trace --stats -d $out/kt5 -o kt --output-format csv -- python tools/bench_configs.py c5 > $out/c5.log 2>&1
cp $(find $out/kt5 -name "*kernel_stats.csv" | head -1) $out/c5_kernel_stats.csv; rm -rf $out/kt5
python tools/bench_train.py --kernels > $out/train_unrolled_pnp.log 2>&1
python tools/bench_c5.py > $out/c5_steady_ab.log 2>&1
python tools/bench_c5.py bf16 >> $out/c5_steady_ab.log 2>&1
tools/c5_timeline.sh > /dev/null 2>&1; cp gpurun_out/c5tl/timeline.txt $out/c5_timeline.txt
(python tools/bench_c4.py 4 32; DPX_CG_WAVE_FFT=2 python tools/bench_c4.py 4 32; DPX_CONV_TILE_ROWS=8 python tools/bench_c4.py 4) > $out/c4_ab.log 2>&1
python tools/bench_shapes.py 8x3x1024x1024 8x3x768x1024 8x3x768x768 8x3x1024x768 1x3x768x1024 1x3x768x768 8x3x1000x1000 8x3x720x1280 8x3x640x640 8x3x500x500 4x3x1536x1536 8x3x1080x1920 > $out/plane_sizes.log 2>&1
# planes off the register-radix path: per-kernel event times, second form of the size-generic transforms against the first, merged z / rhs pass
for s in 8x3x1000x1000 8x3x720x1280 8x3x640x640; do python tools/prof_shape.py $s; DPX_GENERIC_INTERLEAVED=0 python tools/prof_shape.py $s; done > $out/generic_planes_kernels.log 2>&1
python tools/bench_methods.py > $out/bench_methods.log 2>&1
python tools/prof_c4.py 4 > $out/c4shard_events.log 2>&1
head -6 $out/kernel_stats.csv; tail -c 400 $out/bench_steps20_warmup5.json; cat $out/plane_sizes.log
