#!/bin/bash
# tuning aid: like build_variant_fast.sh for the split-bf16 convolution (recompiles dpx_conv_bf16.hip only)
set -e
name=$1; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
out="$ROOT/delta-prox_amd/lib/variants"; obj="$ROOT/delta-prox_amd/build/varc_$name"
mkdir -p "$out" "$obj"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -c "$ROOT/delta-prox_amd/csrc/dpx_conv_bf16.hip" -o "$obj/dpx_conv_bf16.hip.o"
others=$(ls "$ROOT"/delta-prox_amd/build/*.hip.o | grep -v "dpx_conv_bf16.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libdpx_$name.so" "$obj"/*.o $others
echo "$out/libdpx_$name.so"
