cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmc/p$i -o p --output-format csv -- python bench.py --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/pmc/log$i.txt 2>&1
done
python tools/pmc_summary.py gpurun_out/pmc/summary.json gpurun_out/pmc/p1 gpurun_out/pmc/p2 gpurun_out/pmc/p3 > /dev/null 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/pmc/summary.json'))
for k,e in d.items():
    if 'iter_rows' in k or 'k_cols_p2<1024, 64, 8, 2' in k:
        print(k, e['launches_sampled'])
        for n,v in sorted(e['counters'].items()): print(f"   {n:28s} {v:16.1f}")
PY
