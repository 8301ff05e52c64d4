#!/bin/sh
# TEST-ONLY: the device header under MemorySanitizer on the CPU -- reads of locals (on the device: registers), LDS or work area before their first write.
# Builds the emulation as an executable with ROCm's clang (-fsanitize=memory, origins tracked) and runs the emulation tests of every kernel family
# through it (tests/test_kernel_emulation.py: MsanProxy).  Opt-in from pytest: RUN_SANITIZERS=1 python -m pytest tests/test_kernel_emulation.py -k msan
set -e
here="$(cd "$(dirname "$0")" && pwd)"
out=${1:-/tmp/emu_msan}
OUT="$out" "$here/build_msan.sh"
export MSD_EMU_MSAN="$out"
cd "$here/../.."
exec python3 -m pytest tests/test_kernel_emulation.py tests/test_shooting_integrators.py tests/test_restoration.py tests/test_watchdog.py -q -x -k "emulated and not sanitizers" ${PYTEST_ARGS}
