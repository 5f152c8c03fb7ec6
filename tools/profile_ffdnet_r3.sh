#!/bin/bash
# GPU box: matrix-core evidence for the denoiser's DEFAULT arithmetic (split-f16, k_conv3x3_bf16<..., 3>) and split-bf16 (<..., 6>):
# separate rocprofv3 --pmc passes over tools/bench_ffdnet_modes.py (FFDNet-colour forward, 2 x 3 x 1024 x 1024, all four modes).
#   usage: tools/profile_ffdnet_r3.sh <tag>     ->  gpurun_out/ffd_<tag>/ffdnet_pmc.json (+ kernel_stats.csv of the same command)
tag=${1:-r3}
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
out=gpurun_out/ffd_$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python tools/bench_ffdnet_modes.py 2 > $out/bench_ffdnet_modes.log 2> $out/rocprof.err
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv; rm -rf $out/kt
grep "TFLOP" $out/bench_ffdnet_modes.log
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $out/p$i -o p --output-format csv -- python tools/bench_ffdnet_modes.py 2 > /dev/null 2> $out/pmc$i.err
done
python tools/pmc_summary.py $out/pmc_all.json $out/p1 $out/p2 $out/p3 $out/p4 > /dev/null 2>&1
python tools/pmc_ffdnet.py $out/pmc_all.json $out/kernel_stats.csv $out/ffdnet_pmc.json
rm -rf $out/p1 $out/p2 $out/p3 $out/p4
