/* TEST-ONLY: host emulation, kernel family "static" (see emu_common.h) */
#include "emu_common.h"

bool emu_run_static(int NT, int SPT, const EmuArgs &a)
{
    if (NT == 64 && SPT == 1) { EMU_CALL(64, 1, false); return true; }
    if (NT == 64 && SPT == 2) { EMU_CALL(64, 2, false); return true; }
    if (NT == 128 && SPT == 1) { EMU_CALL(128, 1, false); return true; }
    if (NT == 128 && SPT == 2) { EMU_CALL(128, 2, false); return true; }
    if (NT == 192 && SPT == 2) { EMU_CALL(192, 2, false); return true; }
    if (NT == 256 && SPT == 2) { EMU_CALL(256, 2, false); return true; }
    if (NT == 192 && SPT == 3) { EMU_CALL(192, 3, false); return true; }
    if (NT == 320 && SPT == 2) {
        /* a first-pass kernel, followed up by the streamed kernel (msd_api.hip: make_plan) */
        EmuArgs b = a;
        std::vector<int> follow(msd::FOLLOW_HDR + 2*(size_t)a.nscen, 0);
        b.P.follow = follow.data();
        { const EmuArgs &a = b; EMU_CALL(320, 2, 0, false, false, 0, 1); }
        b.P.list = follow.data(); b.P.follow = nullptr;
        { const EmuArgs &a = b; EMU_CALL(128, 5, 0, true); }
        return true;
    }
    return false;
}
