/* solve-kernel instantiations for the dynamic loss model(s); see msd_geometry.hpp */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "msd_geometry.hpp"

namespace msd {
/* full: FULL_BOTH / FULL_RG -- the structure of the reference's rolling stock compiled in (every row on, power rows two-sided, energy objective; with /
 * without the pneumatic brake), where that instantiation exists (msd_kernels_dynamic2.hip, msd_kernels_dynamic3.hip); round 6: 438 k -> 508 k solves/s on
 * the figure-5 batch at N = 100.  The streamed kernel of the family follows up either way */
Geometry pick_geometry_dynamic(int N, int full)
{
    if (full && !tuning().no_full) {
        const Geometry g = full == FULL_BOTH ? pick_geometry_dynamic_full_both(N) : pick_geometry_dynamic_full_rg(N);
        if (g.fn) return g;
    }
    return pick_geometry_t<LOSS_TABLE>(N);
}
}
