#!/bin/bash
# tuning aid: build libdpx_hip with extra compile flags into delta-prox_amd/lib/variants/libdpx_<name>.so (select with DPX_LIB=...)
set -e
name=$1; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
out="$ROOT/delta-prox_amd/lib/variants"; obj="$ROOT/delta-prox_amd/build/var_$name"
mkdir -p "$out" "$obj"
for s in "$ROOT"/delta-prox_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -c "$s" -o "$obj/$(basename $s).o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libdpx_$name.so" "$obj"/*.o
echo "$out/libdpx_$name.so"
