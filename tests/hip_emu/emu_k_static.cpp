/* TEST-ONLY: host emulation, kernel family "static" (see emu_common.h) */
#include "emu_common.h"

bool emu_run_static(int NT, int SPT, const EmuArgs &a, int full)
{
    /* the time-optimal problem with the structure of the rolling stock compiled in (msd_kernels_time.hip / time2.hip) */
    if (full == msd::FULL_TIME_RG && NT == 64 && SPT == 1) { run_first_and_follow<64, 1, 0, false, false, msd::FULL_TIME_RG>(a); return true; }
    if (full == msd::FULL_TIME_RG && NT == 64 && SPT == 2) { run_first_and_follow<64, 2, 0, false, false, msd::FULL_TIME_RG>(a); return true; }
    if (full == msd::FULL_TIME_BOTH && NT == 64 && SPT == 1) { run_first_and_follow<64, 1, 0, false, false, msd::FULL_TIME_BOTH>(a); return true; }
    if (full == msd::FULL_TIME_BOTH && NT == 64 && SPT == 2) { run_first_and_follow<64, 2, 0, false, false, msd::FULL_TIME_BOTH>(a); return true; }
    if (NT == 64 && SPT == 1) { run_first_and_follow<64, 1, 0, false, false, 0>(a); return true; }
    if (NT == 64 && SPT == 2) { run_first_and_follow<64, 2, 0, false, false, 0>(a); return true; }
    if (NT == 128 && SPT == 1) { run_first_and_follow<128, 1, 0, false, false, 0>(a); return true; }
    if (NT == 128 && SPT == 2) { run_first_and_follow<128, 2, 0, false, false, 0>(a); return true; }
    if (NT == 192 && SPT == 2) { run_first_and_follow<192, 2, 0, false, false, 0>(a); return true; }
    if (NT == 256 && SPT == 2) { run_first_and_follow<256, 2, 0, false, false, 0>(a); return true; }
    if (NT == 192 && SPT == 3) { run_first_and_follow<192, 3, 0, false, false, 0>(a); return true; }
    if (NT == 320 && SPT == 2) { run_first_and_follow<320, 2, 0, false, false, 0>(a); return true; }
    return false;
}
