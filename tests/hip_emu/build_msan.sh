#!/bin/sh
# TEST-ONLY: the host emulation of the device header as an executable under MemorySanitizer (ROCm's clang; see emu_msan_main.cpp).
# One translation unit per kernel family, in parallel.  Output: OUT=<path> (default /tmp/emu_msan).
set -e
cd "$(dirname "$0")"
CXX=${CLANGXX:-/opt/rocm/lib/llvm/bin/clang++}
FLAGS="-std=c++17 -O1 -g -fsanitize=memory -fsanitize-memory-track-origins=2 -fno-omit-frame-pointer -pthread -ffp-contract=off -I. $EXTRA"
OUT=${OUT:-/tmp/emu_msan}
OBJ=${OBJDIR:-/tmp/msd_emu_msan_obj}
mkdir -p "$OBJ"
pids=""
for u in emu_msan_main emu_driver emu_k_static emu_k_full emu_k_dynamic emu_k_general emu_k_intloss emu_k_intloss_table emu_k_stream; do
  newest=$(ls -t $u.cpp emu_common.h hip/hip_runtime.h ../../ms-eetc_amd/csrc/*.hpp ../../include/mseetc_hip.h "$OBJ/$u.o" 2>/dev/null | head -1)
  if [ "$newest" != "$OBJ/$u.o" ] || [ -n "$FORCE" ]; then
    $CXX $FLAGS -c -o "$OBJ/$u.o" $u.cpp &
    pids="$pids $!"
  fi
done
for p in $pids; do wait $p; done
$CXX $FLAGS -o "$OUT" "$OBJ"/emu_msan_main.o "$OBJ"/emu_driver.o "$OBJ"/emu_k_*.o
echo "$OUT"
