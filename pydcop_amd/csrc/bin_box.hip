// bin_box.hip -- the lane-grid kernels for binary / unary factors (bin_box.h) as a translation unit of
// their own: one instantiation per (shape, storage type, sign, word), compiled beside engine.hip.
#define MXS_BIN2_IMPL 1
#include "bin_box.h"

namespace mxs {
template bool launch_factor_bin2<double>(const NaryLaunch&, const SweepArgs<double>&, const NaryDesc*, hipStream_t, const ClassInfo*, int);
template bool launch_factor_bin2<float>(const NaryLaunch&, const SweepArgs<float>&, const NaryDesc*, hipStream_t, const ClassInfo*, int);
}  // namespace mxs
