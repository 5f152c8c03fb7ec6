#!/bin/bash
# GPU box: host-side time stamps of one iters(K=20) call on config 2 (DPX_TRACE_HOST=1), third call of a warm solver
cd $GRAFT_REPO_ROOT
DPX_TRACE_HOST=1 K=20 python tools/hosttime.py 2>&1 | grep -E "dpx host|host return" | tail -45
