/* solve-kernel instantiations for combined options beyond the LDS-resident horizons (round 4): collocation / adaptive shooting integrators
 * together with the dynamic loss model or with integrateLosses on the streamed kernel (stage blocks in device memory), up to 1023 intervals --
 * the reference builds any option set for any N (simulations/table3.py:34 sweeps N to 5000); see msd_kernels_compose.hip, msd_kernels_stream3.hip */
#include <hip/hip_runtime.h>

#include "msd_geometry.hpp"

namespace msd {
Geometry pick_stream_geometry_general_dynamic(int N) { return N + 1 <= 1024 ? stream_geometry_t<LOSS_TABLE, true, 2>() : Geometry{0, 0, nullptr}; }
Geometry pick_stream_geometry_general_intloss(int N) { return N + 1 <= 1024 ? stream_geometry_t<LOSS_INTEGRATED, true, 2>() : Geometry{0, 0, nullptr}; }
}
