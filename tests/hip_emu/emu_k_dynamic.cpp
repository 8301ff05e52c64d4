/* TEST-ONLY: host emulation, kernel family "dynamic" (see emu_common.h) */
#include "emu_common.h"

bool emu_run_dynamic(int NT, int SPT, const EmuArgs &a)
{
    if (NT == 64 && SPT == 1) { EMU_CALL(64, 1, true); return true; }
    if (NT == 64 && SPT == 2) { EMU_CALL(64, 2, true); return true; }
    if (NT == 128 && SPT == 1) { EMU_CALL(128, 1, true); return true; }
    if (NT == 128 && SPT == 2) { EMU_CALL(128, 2, true); return true; }
    if (NT == 192 && SPT == 2) { EMU_CALL(192, 2, true); return true; }
    return false;
}
