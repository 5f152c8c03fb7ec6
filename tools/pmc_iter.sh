#!/bin/bash
# GPU box: SQ counters (instruction mix, issue / wait cycles) of the two kernels of the two-kernel ADMM iteration, at the headline size and on
# one rank's shard of an 8-way batch split.  Counter passes are rocprofv3 --pmc with --kernel-trace ONLY (the pool's rule).
#   tools/pmc_iter.sh [outdir]        -> <outdir>/summary.json + summary.txt
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
OUT=${1:-gpurun_out/r5/pmc_iter}
mkdir -p $OUT
cat > /tmp/pmc_iter_work.py <<'PY'
import os, sys
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch, dprox as dp, synthetic
os.environ["DPX_CHAINS"] = "1"
for shape in ((8, 3, 1024, 1024), (1, 3, 1024, 1024)):
    gt, b, psf = synthetic.deconv_case(*shape, seed=1)
    bt = torch.from_numpy(b).cuda()
    x = dp.Variable()
    s = dp.compile(dp.sum_squares(dp.conv(x, psf) - bt) + dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1)), method="admm", device="cuda")
    s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=12)
    torch.cuda.synchronize()
PY
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p --output-format csv -- python /tmp/pmc_iter_work.py > $OUT/log$i.txt 2>&1
done
python - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("dpx::", "")
        if not (k.startswith("k_iter_rows_seq<512, 64, 2, true") or k.startswith("k_iter_rows_par<512, 64, 2, true") or k.startswith("k_cols_p2<1024")):
            continue
        key = f"{k} grid={row['Grid_Size']}"
        acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {"launches": max(len(x) for x in v.values()), "counters": {n: sum(x) / len(x) for n, x in v.items()}} for k, v in acc.items()}
json.dump(res, open(out + "/summary.json", "w"), indent=1, sort_keys=True)
with open(out + "/summary.txt", "w") as fh:
    for k, e in sorted(res.items()):
        c = e["counters"]
        print(k, "launches", e["launches"], file=fh)
        for n, v in sorted(c.items()):
            per_wave = v / c["SQ_WAVES"] if c.get("SQ_WAVES") else float("nan")
            print(f"   {n:24s} {v:16.1f}   per wave {per_wave:12.1f}", file=fh)
print(open(out + "/summary.txt").read())
PY
