#!/bin/bash
# TEST INFRASTRUCTURE: the kernel sources compiled for the host (SIMT emulator) WITH AddressSanitizer, run through the
# parity cases -- catches out-of-bounds global reads / writes that happen to produce the right numbers on a GPU
# (it would have caught the weight over-read of layers with < 8 input channels).  Usage: tests/emul/asan/run_asan.sh [case ...]
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"; EMUL="$(dirname "$HERE")"; ROOT="$(cd "$EMUL/../.." && pwd)"
CXX=${EMUL_CXX:-/opt/rocm/lib/llvm/bin/clang++}
OUT=${TMPDIR:-/tmp}/dpx_asan; mkdir -p "$OUT"
for s in "$ROOT"/delta-prox_amd/csrc/*.hip "$EMUL"/emul.cpp; do
  $CXX -x c++ -std=c++17 -O1 -g -fPIC -fsanitize=address -shared-libasan -fno-omit-frame-pointer -Wno-unused-value -Wno-psabi -I "$EMUL" -c "$s" -o "$OUT/$(basename "$s").o" &
done
wait
$CXX -shared -fsanitize=address -shared-libasan -o "$OUT/libdpx_emul_asan.so" "$OUT"/*.o
RT=$(dirname "$($CXX -print-file-name=libclang_rt.asan-x86_64.so)")/libclang_rt.asan-x86_64.so
CASES=${@:-conv2d linops config1 small pow2 csmri sisr doe grads cg ffdnet ffbwd ladmm other bf16hist lsolve unet ffmodes pgd h768 sizes hqs}
for c in $CASES; do
  LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:handle_segv=0 DPX_ASAN_LIB="$OUT/libdpx_emul_asan.so" \
    python "$HERE/run_cases.py" $c 2>&1 | grep -E "^OK|ERROR: AddressSanitizer|SUMMARY|^\s+#[0-9] " | head -12
done
