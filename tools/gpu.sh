#!/bin/bash
# builder's helper: rebuild libdpx_hip.so (the snapshot ships the built .so), then run a command on an MI355X box through gpurun
# usage: tools/gpu.sh <timeout-seconds> '<command>'
set -e
cd "$(dirname "$0")/.."
python __graft_entry__.py > /tmp/dpx_build.log 2>&1 || { tail -30 /tmp/dpx_build.log; exit 1; }
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
