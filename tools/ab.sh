#!/bin/bash
# tuning aid (GPU box): bench.py per-kernel times for each library variant given on the command line
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== $v"
  DPX_LIB=$GRAFT_REPO_ROOT/delta-prox_amd/lib/variants/libdpx_$v.so python bench.py --no-cpu-baseline --no-extra-configs 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'psnr', round(d['psnr_db']['admm50_mean'],3), {k:round(v['avg_us'],1) for k,v in d['kernels'].items() if 'iter' in k or 'cols_p2' in k})"
done
