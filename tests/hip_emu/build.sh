#!/bin/sh
# TEST-ONLY: host emulation build of the device header (optionally with sanitizers: SAN=1)
set -e
cd "$(dirname "$0")"
FLAGS="-O1 -g"
[ -n "$SAN" ] && FLAGS="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer"
g++ -std=c++17 $FLAGS -fPIC -shared -pthread -ffp-contract=off -I. -o libmsd_emu.so emu_driver.cpp
