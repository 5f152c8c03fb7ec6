#!/bin/bash
# tuning aid (GPU box): ms per iteration (K = 20 and 200) of config 2 for stream set-ups of the sub-batch chains
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" ITERS=200 python tools/concurrent_probe.py quick 2>&1 | grep "2 chains, auto\|1 chain"; env "$@" ITERS=20 python tools/concurrent_probe.py quick 2>&1 | grep "2 chains, auto\|1 chain"; }
run X=0
run DPX_CHAIN_ALLSIDE=1
run DPX_CHAIN_PRIO=-1
run DPX_CHAIN_ALLSIDE=1 DPX_CHAIN_PRIO=-1
run DPX_CHAIN_LOCKSTEP=1
