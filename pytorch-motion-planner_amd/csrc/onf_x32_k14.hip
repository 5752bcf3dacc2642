// Instances of csrc/onf_x32_impl.h for 14 input blocks of 16 (its own translation unit: the build compiles the feature
// dimensions in parallel).
#include "onf_x32_impl.h"

namespace nfopp {
namespace x32 {
int launch_nkb14(const OnfKernelArgs& a, hipStream_t stream, int mode, int* grid_out) { return launch_modes<14>(a, stream, mode, grid_out); }
}  // namespace x32
}  // namespace nfopp
