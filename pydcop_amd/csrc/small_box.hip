// small_box.hip -- the lane-group kernels for n-ary factors over small domains (small_box.h) as a translation unit of
// their own: one instantiation per (arity, storage type, sign, word), compiled beside engine.hip.
#define MXS_SMALL_IMPL 1
#include "small_box.h"

namespace mxs {
template bool launch_factor_small<double>(const NaryLaunch&, const SweepArgs<double>&, const NaryDesc*, hipStream_t);
template bool launch_factor_small<float>(const NaryLaunch&, const SweepArgs<float>&, const NaryDesc*, hipStream_t);
}  // namespace mxs
