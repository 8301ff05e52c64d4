/* integrateLosses with a loss table (msd_lossint_table.hpp), the streamed kernels: first pass and follow-up kernel (restoration phase, watchdog procedure,
 * second attempt) for horizons of up to 1023 intervals; the follow-up kernel also serves the LDS-resident first-pass kernels of msd_kernels_intloss_table.hip */
#include <hip/hip_runtime.h>

#include "msd_geometry.hpp"

namespace msd {
Geometry pick_stream_geometry_intloss_table(int N) { return N + 1 <= 1024 ? stream_geometry_t<LOSS_INTEGRATED_TABLE, false, 2>() : Geometry{0, 0, nullptr}; }
}
