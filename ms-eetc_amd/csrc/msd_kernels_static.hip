/* solve-kernel instantiations for the static loss model(s); see msd_geometry.hpp */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "msd_geometry.hpp"

namespace msd {
Geometry pick_geometry_static(int N, int full)
{
    if (full_energy(full)) { const Geometry g = (full == FULL_RG) ? pick_geometry_full_rg(N) : pick_geometry_full(N); if (g.fn) return g; }
    /* the time-optimal problem on the reference's rolling stock: first-pass kernels of the general iteration with that structure compiled in (msd_kernels_time.hip,
     * msd_kernels_time2.hip), the streamed kernel of the family follows up */
    if (full_time(full) && !tuning().no_full) { const Geometry g = (full == FULL_TIME_RG) ? pick_geometry_time_rg(N) : pick_geometry_time_both(N); if (g.fn) return g; }
    return pick_geometry_t<LOSS_STATIC>(N);
}
}
