#!/bin/bash
# tuning aid: like build_variant.sh but recompiles ONE source file with the extra flags and links it with the main build's other objects
# (run `python __graft_entry__.py` first):  tools/build_variant_one.sh <name> <source stem, e.g. dpx_wgrad_c8> [flags ...]
set -e
name=$1; stem=$2; shift 2
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
out="$ROOT/delta-prox_amd/lib/variants"; obj="$ROOT/delta-prox_amd/build/var1_$name"
mkdir -p "$out" "$obj"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -c "$ROOT/delta-prox_amd/csrc/$stem.hip" -o "$obj/$stem.hip.o"
others=$(ls "$ROOT"/delta-prox_amd/build/*.hip.o | grep -v "/$stem.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libdpx_$name.so" "$obj/$stem.hip.o" $others
echo "$out/libdpx_$name.so"
