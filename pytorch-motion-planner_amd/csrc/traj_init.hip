// Batched trajectory initialisation on the device (the step BEFORE `step()`).
//
// Replaces TrajectoryInitializer.initialize_trajectory / initialize_angle /
// initialize_angle_with_trajectory_direction (nfop/trajectory_initializer.py:12-45) for a whole batch: xy on the
// straight segment start -> goal, theta along the wrapped shortest rotation, optionally pulled towards the travel
// direction with a 0 -> 1 -> 0 ramp.  `torch.linspace` on CPU evaluates start + i*step for the first half and
// end - (steps-1-i)*step for the second, each as one fused multiply-add with an fp32 step; the same expression is used
// here so the waypoints are bit-identical to the reference's (headings of the directed variant: atan2 rounding only).
#include "common.h"

namespace nfopp {

constexpr int TI_THREADS = 256;

__device__ __forceinline__ float linspace_at(float s, float e, int steps, int i) {
  if (steps <= 1) return s;
  const float step = (e - s) / (float)(steps - 1);   // IEEE division (hipcc default)
  return i < steps / 2 ? fmaf(step, (float)i, s) : fmaf(-step, (float)(steps - 1 - i), e);
}

struct InitArgs {
  const float* start; const float* goal;   // [B, D]
  int n, directed;
  float* traj;                             // [B, N, D]
};

template <int D>
__global__ __launch_bounds__(TI_THREADS) void traj_init_kernel(const InitArgs a) {
  const long long b = blockIdx.x;
  const int N = a.n, steps = N + 2;
  float s[D], g[D];
#pragma unroll
  for (int d = 0; d < D; ++d) { s[d] = a.start[b * D + d]; g[d] = a.goal[b * D + d]; }
  float goal_angle = 0.f;
  if (D == 3) goal_angle = wrap_angle(g[2] - s[2]) + s[2];
  float* out = a.traj + b * (long long)N * D;
  for (int i = threadIdx.x; i < N; i += TI_THREADS) {
    out[i * D + 0] = linspace_at(s[0], g[0], steps, i + 1);
    out[i * D + 1] = linspace_at(s[1], g[1], steps, i + 1);
    if (D == 3) {
      float th = linspace_at(s[2], goal_angle, steps, i + 1);
      if (a.directed) {
        // central difference over the FULL path (start, waypoints, goal): neighbours i and i+2 of the full index
        const float x0 = i == 0 ? s[0] : linspace_at(s[0], g[0], steps, i);
        const float y0 = i == 0 ? s[1] : linspace_at(s[1], g[1], steps, i);
        const float x1 = i == N - 1 ? g[0] : linspace_at(s[0], g[0], steps, i + 2);
        const float y1 = i == N - 1 ? g[1] : linspace_at(s[1], g[1], steps, i + 2);
        const float heading = atan2f(y1 - y0, x1 - x0);
        const int h = N / 2;
        const float w = i < h ? linspace_at(0.f, 1.f, h, i) : linspace_at(1.f, 0.f, (N + 1) / 2, i - h);
        th = add_mul_unfused(th, wrap_angle(heading - th), w);
      }
      out[i * D + 2] = th;
    }
  }
}

}  // namespace nfopp

using namespace nfopp;

extern "C" int nfopp_init_trajectories(const float* start_dev, const float* goal_dev, int64_t batch,
                                       int32_t n_waypoints, int32_t dim, int32_t angles_with_direction,
                                       float* traj_dev, void* stream) {
  NFOPP_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
  NFOPP_REQUIRE(batch >= 0 && batch <= 0x7fffffffLL && n_waypoints >= 1, "bad sizes");
  NFOPP_REQUIRE(!(angles_with_direction && dim != 3), "heading initialisation needs SE(2) trajectories (dim 3)");
  if (batch == 0) return NFOPP_OK;
  NFOPP_REQUIRE(start_dev && goal_dev && traj_dev, "null device pointer");
  InitArgs a;
  a.start = start_dev; a.goal = goal_dev; a.n = n_waypoints; a.directed = angles_with_direction ? 1 : 0;
  a.traj = traj_dev;
  if (dim == 3) hipLaunchKernelGGL(traj_init_kernel<3>, dim3((unsigned)batch), dim3(TI_THREADS), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(traj_init_kernel<2>, dim3((unsigned)batch), dim3(TI_THREADS), 0, (hipStream_t)stream, a);
  NFOPP_HIP(hipGetLastError());
  return NFOPP_OK;
}
