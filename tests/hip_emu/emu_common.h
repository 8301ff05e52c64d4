/*
 * TEST-ONLY: shared part of the host emulation (see emu_driver.cpp).  run_blocks<...> runs one instantiation of msd::solve_kernel on host
 * threads; the instantiations are spread over one translation unit per kernel family (emu_k_*.cpp) so that they compile in parallel --
 * an AddressSanitizer build of a single unit took twenty minutes.
 */
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "../../ms-eetc_amd/csrc/msd_kernel.hpp"

/* MemorySanitizer build (tests/hip_emu/build_msan.sh, emu_msan_main.cpp): LDS and work area of an emulated workgroup start as uninitialised memory, so a
 * read of shared memory before its first write is reported where it decides something -- with the origin of the value -- instead of showing as a NaN */
#if defined(__has_feature)
#if __has_feature(memory_sanitizer)
#include <sanitizer/msan_interface.h>
#define EMU_POISON_SHARED(p, n) __msan_poison((p), (n))
#endif
#endif
#ifndef EMU_POISON_SHARED
#define EMU_POISON_SHARED(p, n) ((void)0)
#endif


template <int NT, int SPT, int DYN, bool STREAM = false, bool GEN = false, int FULL = 0, int PART = 0, bool SLDS = false, bool SOCK = false>
void run_blocks(msd::DevProb P, int nscen, const double *scen, const double *ovr, double *z, double *lam, double *stats, double *hist, int cap)
{
    for (int b = 0; b < nscen; b++) {
        emu_block blk;
        blk.nthreads = NT;
        pthread_barrier_init(&blk.bar, nullptr, NT);
        std::vector<double> shfl(NT), xch((size_t)NT*EMU_XCH), lds(STREAM ? msd::lds_doubles_stream() : msd::lds_doubles(P.N, NT*SPT, DYN != 0, (msd::full_energy(FULL) && DYN == 0 && !GEN) ? msd::XCH_FAST : msd::XCH_GENERAL,
                                                                                                   (msd::full_energy(FULL) && DYN == 0 && !GEN && NT == 64) ? 0 : msd::RED_DOUBLES) + (STREAM ? 0 : msd::coop_doubles(NT, GEN)) + (SLDS ? msd::STATIC_FIELDS*NT*SPT : 0));      /* (exactly the LDS the launch code allocates: msd_geometry.hpp) */
        /* EMU_POISON=1 (environment): LDS and work area start as NaN instead of zero -- a read of shared memory before its first write, which on the
         * device sees whatever the kernel before left there, then shows in the results */
        const char *poison = getenv("EMU_POISON");
        if (poison && *poison == '1') std::fill(lds.begin(), lds.end(), std::nan(""));
        EMU_POISON_SHARED(lds.data(), 8*lds.size());
        blk.shfl = shfl.data(); blk.xch = xch.data(); blk.lds = lds.data();
        std::vector<double> work((STREAM ? msd::stream_doubles(P.N, NT*SPT, DYN != 0) : msd::work_doubles(NT*SPT))*(size_t)nscen);
        if (poison && *poison == '1') std::fill(work.begin(), work.end(), std::nan(""));
        EMU_POISON_SHARED(work.data(), 8*work.size());
        std::vector<std::thread> th;
        for (int t = 0; t < NT; t++)
            th.emplace_back([&, t]() {
                threadIdx = {(unsigned)t, 0, 0}; blockIdx = {(unsigned)b, 0, 0}; blockDim = {(unsigned)NT, 1, 1}; gridDim = {(unsigned)nscen, 1, 1};
                emu_blk = &blk;
                msd::solve_kernel<NT, SPT, 1, DYN, STREAM, GEN, FULL, PART, SLDS, SOCK>(P, nscen, scen, ovr, z, lam, stats, hist, cap, work.data());
            });
        for (auto &t : th) t.join();
        pthread_barrier_destroy(&blk.bar);
    }
}


/* family dispatchers (emu_k_*.cpp): false when the family has no instantiation for (NT, SPT) */
struct EmuArgs { msd::DevProb P; int nscen; const double *scen, *ovr; double *z, *lam, *stats, *hist; int cap; };
bool emu_run_static(int NT, int SPT, const EmuArgs &a, int full = 0);
bool emu_run_full(int NT, int SPT, const EmuArgs &a, int kind);      /* kind: msd::FULL_BOTH or msd::FULL_RG */
bool emu_run_dynamic(int NT, int SPT, const EmuArgs &a, int full = 0);
bool emu_run_general(int NT, int SPT, const EmuArgs &a);
bool emu_run_intloss(int NT, int SPT, const EmuArgs &a);
bool emu_run_general_intloss(int NT, int SPT, const EmuArgs &a);
bool emu_run_intloss_table(int NT, int SPT, const EmuArgs &a);
bool emu_run_stream(const EmuArgs &a, int full = 0);
#define EMU_CALL(...) run_blocks<__VA_ARGS__>(a.P, a.nscen, a.scen, a.ovr, a.z, a.lam, a.stats, a.hist, a.cap)

/* a split solve of a family without LDS-resident follow-up kernels, like msd_api.hip: launch_plan does it: the first pass (PART = 1: the general iteration
 * without the cold paths) + the streamed follow-up kernel of the family -- restoration phase, watchdog procedure, second attempt -- over the list the first
 * pass left (a stand-in geometry the emulation can afford: 128 x 5) */
template <int NT, int SPT, int DYN, bool STREAM, bool GEN, int FULL> void run_first_and_follow(EmuArgs a)
{
    std::vector<int> follow(msd::FOLLOW_HDR + 2*(size_t)a.nscen, 0);
    a.P.follow = follow.data();
    EMU_CALL(NT, SPT, DYN, STREAM, GEN, FULL, 1);
    a.P.list = follow.data(); a.P.follow = nullptr;
    EMU_CALL(128, 5, DYN, true, GEN, 0, 2);
}
