/* TEST-ONLY: msd::last_interval (ms-eetc_amd/csrc/msd_kernel.hpp) on the host, for tests/test_kernel_units.py: reads a stage block (31 doubles), the terminal
 * value function (Ptt, pt) and the pneumatic-brake flag from stdin, prints ok, Pn[6], pvn[3], K[8], KS[4], LG[7]. */
#include "../emu_common.h"
thread_local emu_dim3 threadIdx, blockIdx, blockDim, gridDim;
thread_local emu_block *emu_blk;
int main()
{
    double s[40] = {0}, Ptt, pt; int pn;
    for (int k = 0; k < 31; k++) if (scanf("%lf", &s[k]) != 1) return 1;
    if (scanf("%lf %lf %d", &Ptt, &pt, &pn) != 3) return 1;
    double Pn[6], pvn[3], K[8], KS[4], LG[7];
    const bool ok = msd::last_interval<0>(s, Ptt, pt, pn != 0, Pn, pvn, K, KS, LG);
    printf("%d\n", ok ? 1 : 0);
    for (int k = 0; k < 6; k++) printf("%.17g ", Pn[k]); printf("\n");
    for (int k = 0; k < 3; k++) printf("%.17g ", pvn[k]); printf("\n");
    for (int k = 0; k < 8; k++) printf("%.17g ", K[k]); printf("\n");
    for (int k = 0; k < 4; k++) printf("%.17g ", KS[k]); printf("\n");
    for (int k = 0; k < 7; k++) printf("%.17g ", LG[k]); printf("\n");
    return 0;
}
