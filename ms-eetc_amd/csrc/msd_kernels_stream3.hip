/* solve-kernel instantiations of the other transcriptions for horizons beyond the LDS-resident kernels (stage blocks in device memory),
 * up to 1023 intervals; see msd_geometry.hpp */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "msd_geometry.hpp"

namespace msd {
Geometry pick_stream_geometry_dynamic(int N) { return N + 1 <= 1024 ? stream_geometry_t<LOSS_TABLE, false, 2>() : Geometry{0, 0, nullptr}; }
Geometry pick_stream_geometry_general(int N) { return N + 1 <= 1024 ? stream_geometry_t<LOSS_STATIC, true, 2>() : Geometry{0, 0, nullptr}; }
Geometry pick_stream_geometry_intloss(int N) { return N + 1 <= 1024 ? stream_geometry_t<LOSS_INTEGRATED, false, 2>() : Geometry{0, 0, nullptr}; }
}
