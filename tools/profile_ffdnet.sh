#!/bin/bash
# GPU box: the denoiser evidence DESIGN.md quotes for config 3 (MFMA-bound): usage tools/profile_ffdnet.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of tools/bench_ffdnet.py (8x3x1024x1024 FFDNet-color forward)  -> kernel_stats.csv
#   2. separate --pmc passes: matrix-core busy cycles / instruction counts / LDS conflicts              -> pmc.json
tag=${1:-x}
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
out=gpurun_out/ffd_$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python tools/bench_ffdnet.py 8 > $out/bench_ffdnet.log 2> $out/rocprof.err
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv; rm -rf $out/kt
grep "TFLOP" $out/bench_ffdnet.log
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $out/p$i -o p --output-format csv -- python tools/bench_ffdnet.py 1 > /dev/null 2> $out/pmc$i.err
done
python tools/pmc_summary.py $out/pmc_all.json $out/p1 $out/p2 > /dev/null 2>&1
python - <<PY
import json
d = json.load(open("$out/pmc_all.json"))
keep = {k: e for k, e in d.items() if "conv3x3_mfma" in k}
for k, e in keep.items():
    c = e["counters"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
        # SQ_BUSY_CYCLES is summed over the 32 XCD-SEs' SQs x ... ; GRBM_GUI_ACTIVE = GPU-busy cycles of the launch.
        # matrix-core utilisation = busy cycles / (GPU-busy cycles x 1024 SIMDs)
        if c.get("GRBM_GUI_ACTIVE"):
            e["mfma_busy_frac_of_simd_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 1024.0)
    print(k[:90], e["launches_sampled"], {n: round(v, 1) for n, v in sorted(c.items())}, e.get("mfma_busy_frac_of_simd_cycles"))
json.dump(keep, open("$out/pmc.json", "w"), indent=1)
PY
rm -rf $out/p1 $out/p2
