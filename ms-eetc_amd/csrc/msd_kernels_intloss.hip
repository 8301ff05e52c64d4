/* solve-kernel instantiations for integrateLosses (loss slacks from the integrated loss power, msd_lossint.hpp); see msd_geometry.hpp */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "msd_geometry.hpp"

namespace msd {
Geometry pick_geometry_intloss(int N, bool full)
{
    if (full) { const Geometry g = pick_geometry_intloss_full(N); if (g.fn) return g; }
    return pick_geometry_t<LOSS_INTEGRATED>(N);
}
}
