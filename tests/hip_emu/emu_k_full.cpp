/* TEST-ONLY: host emulation, kernel family "full" (see emu_common.h) */
#include "emu_common.h"

/* split solves like msd_api.hip launches them: the first pass (fused iteration only), then the follow-up kernel over the list it left
 * (EMU_MONOLITHIC=1: the kernel with everything in it) */
template <int NT, int SPT, int KIND> static void run_split(EmuArgs a)
{
    std::vector<int> follow(msd::FOLLOW_HDR + 2*(size_t)a.nscen, 0);
    a.P.follow = follow.data();
    const bool plain = a.P.guess ? a.P.dualIn != nullptr : a.P.start == MSD_START_PROFILE;      /* like msd_api.hip: launch() */
    /* EMU_SOCK=1: the first-pass kernels with the second-order correction inside the fused iteration (msd_kernels_full4.hip: both brakes, 64 x 1 and the
     * 64 x 2 kernel with the node constants in LDS), which msd_api.hip launches for the re-solves of the shrinking-horizon loop and for a handle whose
     * launches have handed corrections over */
    const char *sock = getenv("EMU_SOCK");
    if (plain && sock && *sock == '1' && KIND == msd::FULL_BOTH && NT == 64) {
        if (SPT == 1) EMU_CALL(64, 1, false, false, false, msd::FULL_BOTH, 1, false, true); else EMU_CALL(64, 2, false, false, false, msd::FULL_BOTH, 1, true, true);
    } else
    if (plain) EMU_CALL(NT, SPT, false, false, false, KIND, 1); else EMU_CALL(NT, SPT, false, false, false, KIND, 3);
    /* the follow-up kernel of the one-node-per-lane geometry is the two-nodes-per-lane one (msd_api.hip: make_plan) */
    const char *nofollow = getenv("EMU_NO_FOLLOW");      /* EMU_NO_FOLLOW=1: the first pass alone (a test that a scenario needs no follow-up kernel) */
    if (nofollow && *nofollow == '1') return;
    a.P.list = follow.data(); a.P.follow = nullptr;
    if (NT == 64 && SPT == 1) EMU_CALL(64, 2, false, false, false, KIND, 2); else EMU_CALL(NT, SPT, false, false, false, KIND, 2);
}

template <int KIND> static bool run_kind(int NT, int SPT, const EmuArgs &a)
{
    const char *mono = getenv("EMU_MONOLITHIC");
    const bool split = !(mono && *mono == '1');
    if (NT == 64 && SPT == 1) { if (split) run_split<64, 1, KIND>(a); else EMU_CALL(64, 1, false, false, false, KIND); return true; }
    if (NT == 64 && SPT == 2) { if (split) run_split<64, 2, KIND>(a); else EMU_CALL(64, 2, false, false, false, KIND); return true; }
    if (NT == 128 && SPT == 2 && KIND == msd::FULL_BOTH) { if (split) run_split<128, 2, KIND>(a); else EMU_CALL(128, 2, false, false, false, KIND); return true; }
    return false;
}

bool emu_run_full(int NT, int SPT, const EmuArgs &a, int kind)
{
    return kind == msd::FULL_RG ? run_kind<msd::FULL_RG>(NT, SPT, a) : run_kind<msd::FULL_BOTH>(NT, SPT, a);
}
