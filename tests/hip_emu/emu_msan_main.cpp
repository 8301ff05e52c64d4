/*
 * TEST-ONLY: the host emulation of the kernels as an executable of its own, for MemorySanitizer (clang -fsanitize=memory; ROCm's clang ships the
 * runtime).  MSan needs an instrumented main program, so the emulation cannot be loaded into the Python interpreter like the ASan/UBSan build is:
 * tests/test_kernel_emulation.py: MsanProxy writes every emu_solve_batch* call of the emulation tests into a case file, runs this program on it and
 * reads the results back -- the tests themselves run unchanged (tests/hip_emu/run_msan.sh).
 *
 * What the run sees that ASan/UBSan and the NaN-poisoned run (EMU_POISON) cannot: a *local* -- on the device a register -- that is read before it is
 * written (round 5: Solver::evs, loaded for every node slot and written for intervals only, made the one-brake follow-up kernels non-deterministic on
 * the device and nowhere else).  LDS and work area start poisoned too (emu_common.h), the outputs are checked for initialised bytes before they are
 * written to the result file.
 *
 *   emu_msan <case file> <result file>
 */
#include "emu_common.h"

#include <cstdint>

#if defined(__has_feature)
#if __has_feature(memory_sanitizer)
#include <sanitizer/msan_interface.h>
#define EMU_MSAN 1
#endif
#endif

extern "C" void emu_set_duals(const double *dual_in, long long stride, double *dual_out);
extern "C" int emu_solve_batch_warm(const msd_problem_desc *d, int nscen, const double *scen, const double *ovr, const double *guess, double mu0, double push,
                                   double *z, double *lam, double *stats, double *hist, int cap);

static void rd(FILE *f, void *p, size_t n) { if (n && fread(p, 1, n, f) != n) { fprintf(stderr, "emu_msan: short case file\n"); exit(3); } }
static std::vector<double> rdv(FILE *f, size_t n) { std::vector<double> v(n); rd(f, v.data(), 8*n); return v; }

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: emu_msan <case file> <result file>\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    /* header: magic, sizeof(desc), nscen, cap, has_ovr, has_guess, dual_in records (0: none), has_dual_out, dual stride, loss doubles, coll doubles, dual doubles per record */
    int64_t h[12];
    rd(f, h, sizeof h);
    if (h[0] != 0x4d53414e || h[1] != (int64_t)sizeof(msd_problem_desc)) { fprintf(stderr, "emu_msan: not a case file of this ABI\n"); return 3; }
    const int nscen = (int)h[2], cap = (int)h[3];
    double mp[2];
    rd(f, mp, sizeof mp);
    msd_problem_desc d;
    rd(f, &d, sizeof d);
    const int N = d.num_intervals, nz = (4 + d.with_pn_brake)*N + 2, rpi = (d.has_power_rows ? 2 : 0) + 3 + (d.energy_optimal ? 2 : 0);
    std::vector<double> ds = rdv(f, N), grad = rdv(f, N), curv = rdv(f, N), bmax = rdv(f, N + 1), loss = rdv(f, h[9]), coll = rdv(f, h[10]);
    d.ds = ds.data(); d.grad = grad.data(); d.curv = curv.data(); d.bmax = bmax.data();
    d.loss_table = h[9] ? loss.data() : nullptr; d.coll_tables = h[10] ? coll.data() : nullptr;
    std::vector<double> scen = rdv(f, (size_t)nscen*MSD_SC_COUNT), ovr = rdv(f, h[4] ? (size_t)nscen*MSD_OV_COUNT : 0), guess = rdv(f, h[5] ? (size_t)nscen*nz : 0);
    std::vector<double> dual_in = rdv(f, (size_t)h[6]*h[11]);
    fclose(f);

    /* z and the statistics: uninitialised until the kernel writes them (checked below); the multipliers of g and the history are written in part only
     * (rows of a failed solve, iterations beyond the history's capacity), so they start as zeros like the tests' arrays */
    double *z = (double *)malloc(8*(size_t)nscen*nz), *stats = (double *)malloc(8*(size_t)nscen*MSD_ST_COUNT);
    std::vector<double> lam((size_t)nscen*rpi*N, 0.0), hist((size_t)cap*msd::HIST_COLS, 0.0), dual_out(h[7] ? (size_t)nscen*h[11] : 0, 0.0);
    emu_set_duals(h[6] ? dual_in.data() : nullptr, h[8], h[7] ? dual_out.data() : nullptr);
    const int rc = emu_solve_batch_warm(&d, nscen, scen.data(), h[4] ? ovr.data() : nullptr, h[5] ? guess.data() : nullptr, mp[0], mp[1], z, lam.data(), stats,
                                        cap ? hist.data() : nullptr, cap);
#ifdef EMU_MSAN
    if (rc == 0) {
        /* an uninitialised byte in the results: reported with the origin of the value */
        __msan_check_mem_is_initialized(stats, 8*(size_t)nscen*MSD_ST_COUNT);
        __msan_check_mem_is_initialized(z, 8*(size_t)nscen*nz);
        __msan_check_mem_is_initialized(lam.data(), 8*lam.size());
        __msan_check_mem_is_initialized(hist.data(), 8*hist.size());
        __msan_check_mem_is_initialized(dual_out.data(), 8*dual_out.size());
    }
#endif
    FILE *o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 2; }
    const int64_t r = rc;
    fwrite(&r, 8, 1, o);
    if (rc == 0) {
        fwrite(z, 8, (size_t)nscen*nz, o); fwrite(lam.data(), 8, lam.size(), o); fwrite(stats, 8, (size_t)nscen*MSD_ST_COUNT, o);
        fwrite(hist.data(), 8, hist.size(), o); fwrite(dual_out.data(), 8, dual_out.size(), o);
    }
    fclose(o);
    free(z); free(stats);
    return 0;
}
