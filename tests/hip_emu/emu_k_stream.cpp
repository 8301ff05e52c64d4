/* TEST-ONLY: host emulation, the long-horizon kernel at a thread count the emulation can afford (see emu_common.h) */
#include "emu_common.h"

bool emu_run_stream(const EmuArgs &a) { EMU_CALL(128, 5, false, true); return true; }
