/* TEST-ONLY: host emulation, streamed kernels (see emu_common.h) */
#include "emu_common.h"

bool emu_run_stream(const EmuArgs &a) { run_first_and_follow<128, 5, 0, true, false, 0>(a); return true; }
