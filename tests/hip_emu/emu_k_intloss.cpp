/* TEST-ONLY: host emulation, kernel family "intloss" (see emu_common.h) */
#include "emu_common.h"

bool emu_run_intloss(int NT, int SPT, const EmuArgs &a)
{
    if (NT == 64 && SPT == 1) { EMU_CALL(64, 1, 2); return true; }
    if (NT == 64 && SPT == 2) { EMU_CALL(64, 2, 2); return true; }
    return false;
}
