#!/bin/sh
# TEST-ONLY: the device header under AddressSanitizer + UBSan on the CPU (GPU sanitizers are not available on the pool).
# Builds the emulation with -fsanitize=address,undefined (one translation unit per kernel family, in parallel) and runs the emulation
# tests against it, including the multi-wave geometries (192 x 2, 320 x 2).  Summary: last lines of the output.
# Opt-in from pytest: RUN_SANITIZERS=1 python -m pytest tests/test_kernel_emulation.py -k sanitizers
set -e
here="$(cd "$(dirname "$0")" && pwd)"
out=${1:-/tmp/libmsd_emu_san.so}
SAN=1 OUT="$out" OBJDIR=/tmp/msd_emu_san_obj "$here/build.sh"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export MSD_EMU_LIB="$out"
cd "$here/../.."
exec python3 -m pytest tests/test_kernel_emulation.py tests/test_shooting_integrators.py tests/test_restoration.py tests/test_watchdog.py tests/test_integrated_loss_table.py -q -x -k "emulated and not sanitizers"
