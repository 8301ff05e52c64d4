/* solve-kernel instantiations for long horizons (stage blocks in device memory), up to 2047 intervals; see msd_geometry.hpp */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "msd_geometry.hpp"

namespace msd {
/* full: FULL_BOTH / FULL_RG -- the first pass with the structure of the reference's rolling stock compiled in (msd_kernels_stream5.hip, msd_kernels_stream6.hip;
 * round 6: N = 700 / 1000 with the figure-10 train 9.7 / 11.5 -> 8.2 / 9.8 ms per solve, 21 k / 16 k -> 25 k / 18 k solves/s at 1024 per launch).  The follow-up
 * kernel is the family's general one either way */
Geometry pick_stream_geometry_static(int N, int full)
{
    Geometry g = pick_stream_geometry_short_t<LOSS_STATIC>(N);
    if (!g.fn) g = pick_stream_geometry_static_long(N);
    if (g.fn && full && !tuning().no_full) {
        const KernelFn f = full == FULL_BOTH ? stream_first_pass_full_both(g.SPT) : stream_first_pass_full_rg(g.SPT);
        if (f) g.fn = f;
    }
    return g;
}
}
