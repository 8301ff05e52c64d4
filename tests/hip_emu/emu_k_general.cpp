/* TEST-ONLY: host emulation, kernel family "general" (see emu_common.h) */
#include "emu_common.h"

/* like msd_api.hip launches this family: a first-pass kernel (general iteration without the restoration phase), then the streamed kernel of the
 * family -- which has the phase -- over the list the first pass left (a stand-in geometry the emulation can afford: 128 x 5) */
template <int NT, int SPT> static void run_split(EmuArgs a) { run_first_and_follow<NT, SPT, 0, false, true, 0>(a); }

bool emu_run_general(int NT, int SPT, const EmuArgs &a)
{
    if (NT == 64 && SPT == 1) { run_split<64, 1>(a); return true; }
    if (NT == 64 && SPT == 2) { run_split<64, 2>(a); return true; }
    return false;
}

/* the same integrators with integrateLosses (loss rows from the integrated loss distance, msd_lossint.hpp) */
bool emu_run_general_intloss(int NT, int SPT, const EmuArgs &a)
{
    if (NT == 64 && SPT == 1) { run_first_and_follow<64, 1, 2, false, true, 0>(a); return true; }
    return false;
}
