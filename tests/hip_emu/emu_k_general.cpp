/* TEST-ONLY: host emulation, kernel family "general" (see emu_common.h) */
#include "emu_common.h"

bool emu_run_general(int NT, int SPT, const EmuArgs &a)
{
    if (NT == 64 && SPT == 1) { EMU_CALL(64, 1, false, false, true); return true; }
    if (NT == 64 && SPT == 2) { EMU_CALL(64, 2, false, false, true); return true; }
    return false;
}

/* the same integrators with integrateLosses (loss rows from the integrated loss distance, msd_lossint.hpp) */
bool emu_run_general_intloss(int NT, int SPT, const EmuArgs &a)
{
    if (NT == 64 && SPT == 1) { EMU_CALL(64, 1, 2, false, true); return true; }
    return false;
}
