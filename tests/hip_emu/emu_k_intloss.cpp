/* TEST-ONLY: host emulation, kernel family "integrated losses" (see emu_common.h) */
#include "emu_common.h"

bool emu_run_intloss(int NT, int SPT, const EmuArgs &a)
{
    if (NT == 64 && SPT == 1) { run_first_and_follow<64, 1, 2, false, false, 0>(a); return true; }
    if (NT == 64 && SPT == 2) { run_first_and_follow<64, 2, 2, false, false, 0>(a); return true; }
    return false;
}
