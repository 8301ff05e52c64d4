/* TEST-ONLY: host emulation, kernel family "dynamic loss table" (see emu_common.h) */
#include "emu_common.h"

bool emu_run_dynamic(int NT, int SPT, const EmuArgs &a, int full)
{
    /* the structure of the rolling stock compiled in (msd_kernels_dynamic2.hip / 3.hip), two geometries each */
    if (full == msd::FULL_RG && NT == 64 && SPT == 1) { run_first_and_follow<64, 1, 1, false, false, msd::FULL_RG>(a); return true; }
    if (full == msd::FULL_RG && NT == 128 && SPT == 1) { run_first_and_follow<128, 1, 1, false, false, msd::FULL_RG>(a); return true; }
    if (full == msd::FULL_BOTH && NT == 64 && SPT == 1) { run_first_and_follow<64, 1, 1, false, false, msd::FULL_BOTH>(a); return true; }
    if (full == msd::FULL_BOTH && NT == 128 && SPT == 1) { run_first_and_follow<128, 1, 1, false, false, msd::FULL_BOTH>(a); return true; }
    if (NT == 64 && SPT == 1) { run_first_and_follow<64, 1, 1, false, false, 0>(a); return true; }
    if (NT == 64 && SPT == 2) { run_first_and_follow<64, 2, 1, false, false, 0>(a); return true; }
    if (NT == 128 && SPT == 1) { run_first_and_follow<128, 1, 1, false, false, 0>(a); return true; }
    if (NT == 128 && SPT == 2) { run_first_and_follow<128, 2, 1, false, false, 0>(a); return true; }
    if (NT == 192 && SPT == 2) { run_first_and_follow<192, 2, 1, false, false, 0>(a); return true; }
    return false;
}
