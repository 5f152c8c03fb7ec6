#!/bin/bash
# tuning aid (GPU box): `bench.py --steps 20 --warmup 5` and the 200-step leg for each library variant ("main" = the built library), twice each
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/delta-prox_amd/lib/variants/libdpx_$v.so
  [ "$v" = main ] && lib=$GRAFT_REPO_ROOT/delta-prox_amd/lib/libdpx_hip.so
  DPX_LIB=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('$v', 'K20', round(d['value'],1), 'steady', round(d['steady_state']['it_per_s'],1), 'psnr', round(d['psnr_db']['admm50_mean'],3), 'parity', d.get('parity_rel_l2'), {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})"
done
done
