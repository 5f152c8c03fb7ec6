#!/bin/bash
# tuning aid: like build_variant.sh but recompiles only the two files of the fused iteration (dpx_iter.hip, dpx_fft_pow2.hip)
# and links them with the main build's other objects (run `python __graft_entry__.py` first).
set -e
name=$1; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
out="$ROOT/delta-prox_amd/lib/variants"; obj="$ROOT/delta-prox_amd/build/varf_$name"
mkdir -p "$out" "$obj"
for f in dpx_iter dpx_fft_pow2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -c "$ROOT/delta-prox_amd/csrc/$f.hip" -o "$obj/$f.hip.o" &
done
wait
others=$(ls "$ROOT"/delta-prox_amd/build/*.hip.o | grep -v "dpx_iter.hip.o\|dpx_fft_pow2.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libdpx_$name.so" "$obj"/*.o $others
echo "$out/libdpx_$name.so"
