#!/bin/sh
# TEST-ONLY: host emulation build of the device header, one translation unit per kernel family in parallel
# (optionally with sanitizers: SAN=1; output: OUT=<path>, default libmsd_emu.so here)
set -e
cd "$(dirname "$0")"
FLAGS="-O1 -g $EXTRA"
[ -n "$SAN" ] && FLAGS="-O1 -g $EXTRA -fsanitize=address,undefined -fno-omit-frame-pointer"
OUT=${OUT:-libmsd_emu.so}
OBJ=${OBJDIR:-obj${SAN:+_san}}
mkdir -p "$OBJ"
pids=""
for u in ${UNITS:-emu_driver emu_k_static emu_k_full emu_k_dynamic emu_k_general emu_k_intloss emu_k_intloss_table emu_k_stream}; do   # UNITS="emu_k_stream ...": only these (a quick look at one family; the other objects must be up to date)
  g++ -std=c++17 $FLAGS -fPIC -pthread -ffp-contract=off -I. -c -o "$OBJ/$u.o" $u.cpp &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
g++ -std=c++17 $FLAGS -fPIC -shared -pthread -o "$OUT" "$OBJ"/emu_driver.o "$OBJ"/emu_k_*.o
