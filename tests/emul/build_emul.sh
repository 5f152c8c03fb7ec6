#!/bin/bash
# TEST INFRASTRUCTURE: builds the kernels of delta-prox_amd/csrc for the HOST against the SIMT emulator.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
CXX=${EMUL_CXX:-/opt/rocm/lib/llvm/bin/clang++}
OUT="$HERE/libdpx_emul.so"
STUB="$HERE/librccl_stub.so"
if [ ! -f "$STUB" ] || [ "$HERE/rccl_stub.cpp" -nt "$STUB" ]; then
  $CXX -std=c++17 -O2 -g -fPIC -shared -o "$STUB" "$HERE/rccl_stub.cpp" -lrt -lpthread
  echo "built $STUB"
fi
SRCS=$(ls "$ROOT"/delta-prox_amd/csrc/*.hip)
NEWEST=$(ls -t $SRCS "$ROOT"/delta-prox_amd/csrc/*.h "$ROOT"/include/*.h "$HERE"/emul.cpp "$HERE"/hip/hip_runtime.h | head -1)
if [ -f "$OUT" ] && [ "$OUT" -nt "$NEWEST" ]; then exit 0; fi
OBJS=""
for s in $SRCS "$HERE/emul.cpp"; do
  o="$HERE/obj_$(basename "$s").o"
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ -n "$(find "$ROOT"/delta-prox_amd/csrc/*.h "$ROOT"/include/*.h "$HERE"/hip/hip_runtime.h -newer "$o" 2>/dev/null)" ]; then
    $CXX -x c++ -std=c++17 -O2 -g -fPIC -Wno-unused-value -Wno-psabi -I "$HERE" -c "$s" -o "$o" &
  fi
  OBJS="$OBJS $o"
done
wait
$CXX -shared -o "$OUT" $OBJS
echo "built $OUT"
