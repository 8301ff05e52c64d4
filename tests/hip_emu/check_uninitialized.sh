#!/bin/sh
# TEST-ONLY gate: the compilers' own analysis of reads before writes over every kernel family of the device header (host emulation units).
#   g++   -O2 -Werror=uninitialized -Werror=maybe-uninitialized      (data-flow analysis after inlining: what named Solver::evs in round 5)
#   clang -Werror=uninitialized -Werror=sometimes-uninitialized -Werror=conditional-uninitialized      (ROCm's clang, the device compiler's front end)
# Exit code 0: no finding.  Twenty minutes of CPU (g++ -O2 on the fused kernels); opt-in from pytest: RUN_SANITIZERS=1 ... -k uninitialized
cd "$(dirname "$0")"
OBJ=${OBJDIR:-/tmp/msd_emu_warn_obj}
CLANGXX=${CLANGXX:-/opt/rocm/lib/llvm/bin/clang++}
mkdir -p "$OBJ"
units="emu_driver emu_k_static emu_k_full emu_k_dynamic emu_k_general emu_k_intloss emu_k_intloss_table emu_k_stream"
rc=0
for u in $units; do
  ( g++ -std=c++17 -O2 -fPIC -pthread -ffp-contract=off -I. -Werror=uninitialized -Werror=maybe-uninitialized -c -o "$OBJ/$u.o" $u.cpp > "$OBJ/$u.gcc.log" 2>&1; echo $? > "$OBJ/$u.gcc.rc" ) &
done
for u in $units; do
  $CLANGXX -std=c++17 -fsyntax-only -pthread -I. -Wno-unused-value -Werror=uninitialized -Werror=sometimes-uninitialized -Werror=conditional-uninitialized $u.cpp > "$OBJ/$u.clang.log" 2>&1 || { rc=1; echo "clang: $u"; tail -20 "$OBJ/$u.clang.log"; }
done
wait
for u in $units; do
  if [ "$(cat "$OBJ/$u.gcc.rc")" != 0 ]; then rc=1; echo "g++: $u"; grep -A6 "error" "$OBJ/$u.gcc.log" | head -40; fi
done
[ $rc = 0 ] && echo "no read before write named by g++ -O2 or clang in the $(echo $units | wc -w) emulation units"
exit $rc
