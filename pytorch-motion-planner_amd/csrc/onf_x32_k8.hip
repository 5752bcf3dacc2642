// Instances of csrc/onf_x32_impl.h for 8 input blocks of 16 (its own translation unit: the build compiles the feature
// dimensions in parallel).
#include "onf_x32_impl.h"

namespace nfopp {
namespace x32 {
#ifndef X32_ONLY_NKB14   /* development: compile the F = 208..223 instances only (a quarter of the build time) */
int launch_nkb8(const OnfKernelArgs& a, hipStream_t stream, int mode, int* grid_out) { return launch_modes<8>(a, stream, mode, grid_out); }
#else
int launch_nkb8(const OnfKernelArgs& a, hipStream_t, int, int*) {
  set_error("this development build holds the F = 208..223 instances only (fin = %d)", a.geom.fin);
  return NFOPP_ERR_ARG;
}
#endif
}  // namespace x32
}  // namespace nfopp
