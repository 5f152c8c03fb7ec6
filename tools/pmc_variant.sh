#!/bin/bash
# GPU box: HBM read / write bytes per launch (FETCH_SIZE, WRITE_SIZE passes) for library variants: tools/pmc_variant.sh <name>...
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
for v in "$@"; do
  out=gpurun_out/pmcv_$v; mkdir -p $out
  for c in FETCH_SIZE WRITE_SIZE; do
    DPX_LIB=$GRAFT_REPO_ROOT/delta-prox_amd/lib/variants/libdpx_$v.so rocprofv3 --pmc $c --kernel-trace -d $out/pmc_$c -o p --output-format csv -- python bench.py --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2> $out/pmc_$c.err
  done
  echo "== $v"; python tools/pmc_summary.py $out/pmc_hbm.json $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE | grep "k_cols_p2\|k_iter_rows"
  rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
done
