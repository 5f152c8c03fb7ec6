#!/bin/bash
# GPU box: the measurements DESIGN.md / profiles/ quote for round 6.  usage: tools/profile_round5.sh <tag>  -> gpurun_out/prof_<tag>/
tag=${1:-r6}
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
# issue / wait counters of the iteration's kernels (k_cols_p2, k_iter_rows_seq at 8x3x1024^2; k_iter_rows_par at 1x3x1024^2)
tools/pmc_iter.sh $out/pmc_sq > /dev/null 2>&1; cp $out/pmc_sq/summary.txt $out/cols_rows_sq_counters.txt; cp $out/pmc_sq/summary.json $out/cols_rows_sq_counters.json; rm -rf $out/pmc_sq
# one rank's shard of the batch on 8 GPUs: launch geometries of its two kernels
(for s in 1x3x1024x1024 2x3x1024x1024 1x3x768x1024 1x3x512x512 1x1x256x256; do python tools/small_shard_probe.py $s default iter_rows=1 iter_rows=3 2>&1 | grep -v amdgpu.ids; done; for s in 1x3x1024x1024 1x3x512x512; do python tools/small_shard_probe.py $s nonneg default iter_rows=1 iter_rows=3 2>&1 | grep -v amdgpu.ids; done) > $out/shard_probe.log 2>&1
# config 4 shard and config 5: kernel traces of the same library
rocprofv3 --kernel-trace --stats -d $out/kt4 -o kt --output-format csv -- python tools/bench_c4.py 4 > $out/c4shard.log 2>&1
cp $(find $out/kt4 -name "*kernel_stats.csv" | head -1) $out/c4shard_kernel_stats.csv; rm -rf $out/kt4
rocprofv3 --kernel-trace --stats -d $out/kt5 -o kt --output-format csv -- python tools/bench_configs.py c5 > $out/c5.log 2>&1
cp $(find $out/kt5 -name "*kernel_stats.csv" | head -1) $out/c5_kernel_stats.csv; rm -rf $out/kt5
# config 4's shard under the A/B switches of its loop (tag polling / events, one call / staged, folded head and tail) + the timeline of the default
bash tools/c4_ab.sh > $out/c4_loop_ab.log 2>&1; cp gpurun_out/c4b/timeline.txt $out/c4_timeline.txt
python tools/plane_chain_probe.py 2>&1 | grep -v amdgpu.ids > $out/plane_chain_probe.log
python tools/bench_shapes.py 8x3x1024x1024 8x3x768x1024 8x3x768x768 8x3x1024x768 1x3x1024x1024 1x3x768x1024 1x3x768x768 8x3x1000x1000 8x3x720x1280 8x3x640x640 8x3x500x500 4x3x1536x1536 4x3x2048x2048 8x3x1080x1920 > $out/plane_sizes.log 2>&1
for s in 8x3x1000x1000 8x3x720x1280; do python tools/prof_shape.py $s; done > $out/generic_planes_kernels.log 2>&1
python tools/bench_methods.py > $out/bench_methods.log 2>&1
(python tools/bench_c5.py; python tools/bench_c5.py bf16) > $out/c5_steady_ab.log 2>&1
DPX_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-configs > $out/bench_forced_dist_path.json 2>> $out/bench.err
# round 6: Winograd layers against the direct split-f16 ones (accuracy + per-kernel times), the probes behind DESIGN.md sections 9.2 / 9.9
python tools/bench_wino.py 8 > $out/wino_vs_direct.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probe_grid_barrier.hip -o /tmp/probe_grid_barrier.bin && /tmp/probe_grid_barrier.bin > $out/grid_barrier_probe.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_wino_prereq.hip -o /tmp/probe_wino_prereq.bin && /tmp/probe_wino_prereq.bin > $out/wino_prereq_probe.log 2>&1
# round 6: training through the FFDNet prior (split-f16 backward pass, k_wgrad_c8) and the weight-gradient kernel on its own
python tools/bench_train.py --kernels > $out/train_final.json 2>&1
(python tools/bench_wgrad.py; python tools/bench_wgrad.py 4 64 320 320; python tools/bench_wgrad.py 2 16 384 384) 2>&1 | grep mode > $out/wgrad_bench.txt
# matrix-core evidence for the committed convolution kernels
tools/profile_ffdnet_r3.sh $tag > $out/ffdnet_modes.log 2>&1; cp gpurun_out/ffd_$tag/ffdnet_pmc.json $out/ffdnet_pmc.json; cp gpurun_out/ffd_$tag/kernel_stats.csv $out/ffdnet_kernel_stats.csv
python -m pytest tests -m gpu -q > $out/gputests.log 2>&1; cp gpurun_out/parity_achieved_gpu.json $out/parity_achieved_gpu.json
head -6 $out/kernel_stats.csv; tail -c 600 $out/bench_steps20_warmup5.json; cat $out/plane_sizes.log; tail -3 $out/gputests.log
