/* solve-kernel instantiations of the other transcriptions for horizons beyond the LDS-resident kernels (stage blocks in device memory),
 * up to 1023 intervals; see msd_geometry.hpp */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "msd_geometry.hpp"

namespace msd {
Geometry pick_stream_geometry_dynamic(int N) { return N + 1 <= 1024 ? Geometry{512, 2, solve_kernel<512, 2, 2, LOSS_TABLE, true>, true} : Geometry{0, 0, nullptr}; }
Geometry pick_stream_geometry_general(int N) { return N + 1 <= 1024 ? Geometry{512, 2, solve_kernel<512, 2, 2, LOSS_STATIC, true, true>, true} : Geometry{0, 0, nullptr}; }
Geometry pick_stream_geometry_intloss(int N) { return N + 1 <= 1024 ? Geometry{512, 2, solve_kernel<512, 2, 2, LOSS_INTEGRATED, true>, true} : Geometry{0, 0, nullptr}; }
}
