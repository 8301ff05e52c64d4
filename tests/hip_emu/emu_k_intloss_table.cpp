/* TEST-ONLY: host emulation, kernel family "integrateLosses with a loss table" (DYN = LOSS_INTEGRATED_TABLE; see emu_common.h) */
#include "emu_common.h"

bool emu_run_intloss_table(int NT, int SPT, const EmuArgs &a)
{
    if (NT == 64 && SPT == 1) { run_first_and_follow<64, 1, msd::LOSS_INTEGRATED_TABLE, false, false, 0>(a); return true; }
    if (NT == 128 && SPT == 1) { run_first_and_follow<128, 1, msd::LOSS_INTEGRATED_TABLE, false, false, 0>(a); return true; }
    return false;
}
