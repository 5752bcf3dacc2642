// Path evaluation for a batch of trajectories (the step AFTER the planner step in every
// experiment driver of the reference).
//
// Replaces the evaluation loop of scripts/run_bench_mr.py:109-132: every `check_collision_frequence` iterations the path
// is densified, every pose is tested by the ground-truth checker, the path length is measured, the shortest
// collision-free path is kept and a trajectory that is collision-free but no longer improving stops.  The reference
// delegates densification / validity / length to bench-mr (BenchmarkAdapter.cpp:201-208, OMPL `interpolated` +
// PathLengthMetric), which is not available; the semantics here are the generic ones: `sub` equally spaced poses per
// segment (theta along the wrapped difference), Euclidean xy length of the waypoint polyline.  Parity with bench-mr's
// numbers is therefore unpinned; parity with the oracle restatement is exact.
#include "common.h"

namespace nfopp {

constexpr int PE_THREADS = 256;

struct InterpArgs {
  const float* traj; const float* start; const float* goal;
  int n, dim, sub;
  float* poses;    // [B, (N+1)*sub + 1, D]
  float* length;   // [B]
};

template <int D>
__global__ __launch_bounds__(PE_THREADS) void path_interpolate_kernel(const InterpArgs a) {
  __shared__ float red[PE_THREADS / 64];
  const long long b = blockIdx.x;
  const int N = a.n, M = (N + 1) * a.sub + 1;
  const float* tr = a.traj + b * N * D;
  float* out = a.poses + b * M * D;
  auto point = [&](int f, int d) {  // full trajectory index 0..N+1
    return f == 0 ? a.start[b * D + d] : (f == N + 1 ? a.goal[b * D + d] : tr[(f - 1) * D + d]);
  };
  float part = 0.f;
  for (int s = threadIdx.x; s <= N; s += PE_THREADS) {
    float p0[D], p1[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { p0[d] = point(s, d); p1[d] = point(s + 1, d); }
    const float dx = p1[0] - p0[0], dy = p1[1] - p0[1];
    part += sqrtf(dx * dx + dy * dy);
    float dth = 0.f;
    if (D == 3) dth = wrap_angle(p1[2] - p0[2]);
    for (int k = 0; k < a.sub; ++k) {
      const float u = (float)k / (float)a.sub;
      float* o = out + (s * a.sub + k) * D;
      o[0] = p0[0] + u * dx;
      o[1] = p0[1] + u * dy;
      if (D == 3) o[2] = p0[2] + u * dth;
    }
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int d = 0; d < D; ++d) out[(M - 1) * D + d] = a.goal[b * D + d];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < PE_THREADS / 64; ++w) s += red[w];
    a.length[b] = s;
  }
}

struct SelectArgs {
  const float* labels;   // [B, M] collision labels of the densified poses
  const float* length;   // [B]
  const float* traj;     // [B, N, D]
  int m, n, dim;
  float* best_traj; float* best_length;   // [B, N, D], [B] (inf = none yet)
  unsigned char* collides;                // [B] out
  unsigned char* active;                  // [B] in/out (NULL = no early stop)
};

// run_bench_mr.py:118-126: collision-free and shorter -> new best; collision-free but not shorter -> stop optimising
__global__ __launch_bounds__(PE_THREADS) void path_select_kernel(const SelectArgs a) {
  __shared__ int any_hit;
  const long long b = blockIdx.x;
  if (threadIdx.x == 0) any_hit = 0;
  __syncthreads();
  int hit = 0;
  for (int k = threadIdx.x; k < a.m; k += PE_THREADS) hit |= a.labels[b * a.m + k] != 0.0f;
  if (hit) any_hit = 1;   // benign race: every writer stores 1
  __syncthreads();
  const bool collision = any_hit != 0;
  const bool was_active = !a.active || a.active[b];
  const float len = a.length[b], best = a.best_length[b];
  const bool improve = was_active && !collision && len < best;
  __syncthreads();
  if (improve) {
    const int nd = a.n * a.dim;
    for (int k = threadIdx.x; k < nd; k += PE_THREADS) a.best_traj[b * nd + k] = a.traj[b * nd + k];
  }
  if (threadIdx.x == 0) {
    a.collides[b] = collision ? 1 : 0;
    if (improve) a.best_length[b] = len;
    if (a.active && was_active && !collision && !improve) a.active[b] = 0;
  }
}

}  // namespace nfopp

using namespace nfopp;

extern "C" int nfopp_path_interpolate(const float* traj_dev, const float* start_dev, const float* goal_dev,
                                      int64_t batch, int32_t n_waypoints, int32_t dim, int32_t sub, float* poses_dev,
                                      float* length_dev, void* stream) {
  NFOPP_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
  NFOPP_REQUIRE(batch >= 0 && batch <= 0x7fffffffLL && n_waypoints >= 1 && sub >= 1, "bad sizes");
  if (batch == 0) return NFOPP_OK;
  NFOPP_REQUIRE(traj_dev && start_dev && goal_dev && poses_dev && length_dev, "null device pointer");
  InterpArgs a;
  a.traj = traj_dev; a.start = start_dev; a.goal = goal_dev; a.n = n_waypoints; a.dim = dim; a.sub = sub;
  a.poses = poses_dev; a.length = length_dev;
  if (dim == 3) hipLaunchKernelGGL(path_interpolate_kernel<3>, dim3((unsigned)batch), dim3(PE_THREADS), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(path_interpolate_kernel<2>, dim3((unsigned)batch), dim3(PE_THREADS), 0, (hipStream_t)stream, a);
  NFOPP_HIP(hipGetLastError());
  return NFOPP_OK;
}

extern "C" int nfopp_path_select_best(const float* labels_dev, const float* length_dev, const float* traj_dev,
                                      int64_t batch, int32_t poses_per_path, int32_t n_waypoints, int32_t dim,
                                      float* best_traj_dev, float* best_length_dev, uint8_t* collides_dev,
                                      uint8_t* active_dev, void* stream) {
  NFOPP_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
  NFOPP_REQUIRE(batch >= 0 && batch <= 0x7fffffffLL && poses_per_path >= 1 && n_waypoints >= 1, "bad sizes");
  if (batch == 0) return NFOPP_OK;
  NFOPP_REQUIRE(labels_dev && length_dev && traj_dev && best_traj_dev && best_length_dev && collides_dev,
                "null device pointer");
  SelectArgs a;
  a.labels = labels_dev; a.length = length_dev; a.traj = traj_dev; a.m = poses_per_path; a.n = n_waypoints; a.dim = dim;
  a.best_traj = best_traj_dev; a.best_length = best_length_dev; a.collides = collides_dev; a.active = active_dev;
  hipLaunchKernelGGL(path_select_kernel, dim3((unsigned)batch), dim3(PE_THREADS), 0, (hipStream_t)stream, a);
  NFOPP_HIP(hipGetLastError());
  return NFOPP_OK;
}
