cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
mkdir -p gpurun_out/pmcf
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmcf/p$i -o p --output-format csv -- python tools/bench_ffdnet_modes.py 2 > gpurun_out/pmcf/log$i.txt 2>&1
done
python tools/pmc_summary.py gpurun_out/pmcf/summary.json gpurun_out/pmcf/p1 gpurun_out/pmcf/p2 gpurun_out/pmcf/p3 > /dev/null 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/pmcf/summary.json'))
for k,e in d.items():
    if 'conv3x3' in k:
        print(k, e['launches_sampled'])
        for n,v in sorted(e['counters'].items()): print(f"   {n:28s} {v:16.1f}")
PY
tail -3 gpurun_out/pmcf/log1.txt
