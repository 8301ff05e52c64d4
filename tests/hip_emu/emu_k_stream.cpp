/* TEST-ONLY: host emulation, streamed kernels (see emu_common.h) */
#include "emu_common.h"

/* full: the first pass with the structure of the rolling stock compiled in (msd_kernels_stream5.hip / 6.hip) */
bool emu_run_stream(const EmuArgs &a, int full)
{
    if (full == msd::FULL_RG) run_first_and_follow<128, 5, 0, true, false, msd::FULL_RG>(a);
    else if (full == msd::FULL_BOTH) run_first_and_follow<128, 5, 0, true, false, msd::FULL_BOTH>(a);
    else run_first_and_follow<128, 5, 0, true, false, 0>(a);
    return true;
}
