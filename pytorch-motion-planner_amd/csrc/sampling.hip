// On-device sample generation and ground-truth checkers for continuous ONF learning over a batch of trajectories
// (the step BEFORE the ONF fit; HBM/latency-bound helper kernels).
//
// Replaces (reference, host numpy):
//   * `_sample_collision_checker_points`, `_random_intermediate_positions`, `_offset_positions`,
//     `_sample_random_field_points`  nfop/nerf_opt_planner.py:101-120,135-141, nfop/constrained_nerf_opt_planner.py:57-61,173-176
//   * `_resample_collision_positions`  nfop/nerf_opt_planner.py:122-133 -- weighted sampling WITHOUT replacement of the
//     retained pool (np.random.choice(p=w, replace=False)) restated as an exponential race: key = -log(u) / w, keep the
//     `cap` smallest keys (Efraimidis-Spirakis; same distribution as sequential weighted draws)
//   * ground truth: circle / rectangle checkers nfop/collision_checker/*.py and the occupancy-grid checker of
//     notebooks/onf_planner_image_map.ipynb cell 2.
// All draws come from Philox4x32-10 (key = seed, counter = (draw index, trajectory, offset, stream)), so a run is
// reproducible and independent of how trajectories are sharded over GPUs; oracle/nfopp_oracle.py restates it.
#include "common.h"

namespace nfopp {

constexpr int SM_THREADS = 256;

// stream ids of the per-step draws
enum { STREAM_T = 1, STREAM_COURSE = 2, STREAM_FINE = 3, STREAM_FIELD = 4, STREAM_KEY = 5 };

__device__ __forceinline__ float draw_uniform(unsigned long long seed, unsigned long long traj, unsigned int idx,
                                              unsigned long long offset, unsigned int stream) {
  // counter words: c0 = idx, c1 = stream, (c2, c3) = (global trajectory << 24) ^ offset   (offset < 2^24 steps)
  const unsigned long long lo = (unsigned long long)idx | ((unsigned long long)stream << 32);
  const unsigned long long hi = (traj << 24) ^ offset;
  return philox_uniform(seed, lo, hi);
}

// standard normal from two uniforms (Box-Muller); u1 is mapped to (0, 1] so the log is finite
__device__ __forceinline__ float draw_normal(unsigned long long seed, unsigned long long traj, unsigned int idx,
                                             unsigned long long offset, unsigned int stream) {
  const float u1 = 1.0f - draw_uniform(seed, traj, 2 * idx, offset, stream);
  const float u2 = draw_uniform(seed, traj, 2 * idx + 1, offset, stream);
  return sqrtf(-2.0f * logf(u1)) * cosf(NFOPP_TWO_PI_F * u2);
}

struct SampleArgs {
  const float* prev_traj;  // [B, N, D]
  long long batch, traj_index_offset;
  int n, dim, cap, pool_n, n_field;
  float course_sigma, fine_sigma, angle_sigma;
  float bounds[4];
  unsigned long long seed, offset;
  const float* pool; const float* pool_age;   // [B, cap, D], [B, cap]
  float* cand; float* cand_age;               // [B, C, D], [B, C]   C = cap + N - 1
  float* samples;                             // [B, S, D]           S = (N - 1) + cap + n_field
};

template <int D>
__global__ __launch_bounds__(SM_THREADS) void sample_candidates_kernel(const SampleArgs a) {
  const long long b = blockIdx.x;
  const unsigned long long tg = (unsigned long long)(a.traj_index_offset + b);
  const int N = a.n, C = a.cap + N - 1, S = (N - 1) + a.cap + a.n_field;
  const float* tr = a.prev_traj + b * N * D;
  float* cand = a.cand + b * C * D;
  float* cage = a.cand_age + b * C;
  float* smp = a.samples + b * S * D;
  for (int k = threadIdx.x; k < a.pool_n; k += SM_THREADS) {  // retained pool first
#pragma unroll
    for (int d = 0; d < D; ++d) cand[k * D + d] = a.pool[(b * a.cap + k) * D + d];
    cage[k] = a.pool_age[b * a.cap + k];
  }
  for (int j = threadIdx.x; j < N - 1; j += SM_THREADS) {
    const float t = draw_uniform(a.seed, tg, j, a.offset, STREAM_T);
    float pos[D];
#pragma unroll
    for (int d = 0; d < D; ++d) pos[d] = tr[(j + 1) * D + d] * (1.0f - t) + tr[j * D + d] * t;  // nerf:117 (plain lerp)
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const float sc = d < 2 ? a.course_sigma : a.angle_sigma, sf = d < 2 ? a.fine_sigma : a.angle_sigma;
      smp[j * D + d] = pos[d] + draw_normal(a.seed, tg, D * j + d, a.offset, STREAM_COURSE) * sc;
      cand[(a.pool_n + j) * D + d] = pos[d] + draw_normal(a.seed, tg, D * j + d, a.offset, STREAM_FINE) * sf;
    }
    cage[a.pool_n + j] = 0.0f;
  }
  for (int r = threadIdx.x; r < a.n_field; r += SM_THREADS) {  // uniform field samples, theta ~ U[0, 2 pi)
    float* o = smp + ((N - 1) + a.cap + r) * D;
    o[0] = a.bounds[0] + draw_uniform(a.seed, tg, D * r, a.offset, STREAM_FIELD) * (a.bounds[1] - a.bounds[0]);
    o[1] = a.bounds[2] + draw_uniform(a.seed, tg, D * r + 1, a.offset, STREAM_FIELD) * (a.bounds[3] - a.bounds[2]);
    if (D == 3) o[2] = draw_uniform(a.seed, tg, D * r + 2, a.offset, STREAM_FIELD) * NFOPP_TWO_PI_F;
  }
}

struct ResampleArgs {
  long long batch, traj_index_offset;
  int n_cand, cand_stride, cap, dim, sample_stride, sample_off;  // pool slot of samples: samples[b][sample_off + k]
  unsigned long long seed, offset;
  const float* cand; const float* cand_age;
  const float* logits;   // [B, cand_stride, 4] records of the ONF evaluation (logit first)
  float* pool; float* pool_age;
  float* samples;
};

template <int D>
__global__ __launch_bounds__(SM_THREADS) void resample_pool_kernel(const ResampleArgs a) {
  extern __shared__ float sm[];
  const long long b = blockIdx.x;
  const unsigned long long tg = (unsigned long long)(a.traj_index_offset + b);
  const int C = a.n_cand, CS = a.cand_stride;
  int n2 = 1;
  while (n2 < C) n2 <<= 1;
  float* key = sm;
  int* idx = reinterpret_cast<int*>(sm + n2);
  for (int c = threadIdx.x; c < n2; c += SM_THREADS) {
    float k = __builtin_inff();
    if (c < C) {
      const float logit = a.logits[(b * CS + c) * 4];
      const float age = a.cand_age[b * CS + c];
      const float w = (1.0f / (1.0f + expf(-logit))) * expf(-0.03f * age) + 1e-6f;  // nerf:125-126
      const float u = 1.0f - draw_uniform(a.seed, tg, c, a.offset, STREAM_KEY);     // (0, 1]
      k = -logf(u) / w;
    }
    key[c] = k;
    idx[c] = c;
  }
  __syncthreads();
  for (int size = 2; size <= n2; size <<= 1) {      // bitonic sort, ascending keys (ties by index)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < n2 / 2; t += SM_THREADS) {
        const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const float kl = key[lo], kh = key[hi];
        const int il = idx[lo], ih = idx[hi];
        const bool gt = kl > kh || (kl == kh && il > ih);
        if (gt == up) { key[lo] = kh; key[hi] = kl; idx[lo] = ih; idx[hi] = il; }
      }
      __syncthreads();
    }
  }
  for (int k = threadIdx.x; k < a.cap; k += SM_THREADS) {
    const int c = idx[k];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const float v = a.cand[(b * CS + c) * D + d];
      a.pool[(b * a.cap + k) * D + d] = v;
      a.samples[(b * a.sample_stride + a.sample_off + k) * D + d] = v;
    }
    a.pool_age[b * a.cap + k] = a.cand_age[b * CS + c] + 1.0f;  // nerf:132
  }
}

// ---- ground-truth checkers ------------------------------------------------------------------------------------
struct CheckArgs {
  const float* poses; long long n; int dim;
  const float* obstacles; int n_obstacles;
  float radius; float box[4];
  int has_bounds; float bounds[4];
  const unsigned char* grid; int rows, cols; double origin_x, origin_y, cell;
  // circle checker with a cell index over the (cell-sorted) obstacle points
  const int* cell_start; int cells_x, cells_y; float cell_x0, cell_y0, cell_size;
  float* labels;
};

// |obstacle - pose| < radius, ONE arithmetic for every circle kernel (an explicit fma: left to the compiler the two kernels
// contracted dx*dx + dy*dy differently and disagreed on poses within an ulp of a rim)
__device__ __forceinline__ bool closer_than(float dx, float dy, float radius) {
  return sqrtf(__builtin_fmaf(dx, dx, dy * dy)) < radius;
}

__device__ __forceinline__ bool out_of_bounds(const CheckArgs& a, float x, float y) {
  // nfop/collision_checker/collision_checker.py:12-19
  return a.has_bounds && (x > a.bounds[1] || x < a.bounds[0] || y > a.bounds[3] || y < a.bounds[2]);
}

// mode 0: disc robot against a point cloud (circle_collision_checker.py:11-14)
// mode 1: box robot, obstacle points moved into the robot frame (rectangle_collision_checker.py:11-26)
template <int MODE>
__global__ __launch_bounds__(SM_THREADS) void check_points_kernel(const CheckArgs a) {
  __shared__ float ox[SM_THREADS], oy[SM_THREADS];
  const long long p = blockIdx.x * (long long)SM_THREADS + threadIdx.x;
  const bool valid = p < a.n;
  float x = 0.f, y = 0.f, c = 1.f, s = 0.f;
  if (valid) {
    x = a.poses[p * a.dim];
    y = a.poses[p * a.dim + 1];
    if (MODE == 1) { const float th = a.poses[p * a.dim + 2]; c = cosf(th); s = sinf(th); }
  }
  bool hit = false;
  for (int base = 0; base < a.n_obstacles; base += SM_THREADS) {
    __syncthreads();
    if (base + (int)threadIdx.x < a.n_obstacles) {
      ox[threadIdx.x] = a.obstacles[2 * (base + threadIdx.x)];
      oy[threadIdx.x] = a.obstacles[2 * (base + threadIdx.x) + 1];
    }
    __syncthreads();
    const int m = min(SM_THREADS, a.n_obstacles - base);
    for (int k = 0; k < m; ++k) {
      const float dx = ox[k] - x, dy = oy[k] - y;
      if (MODE == 0) {
        hit |= closer_than(dx, dy, a.radius);
      } else {
        const float rx = c * dx + s * dy, ry = -s * dx + c * dy;
        hit |= rx > a.box[0] && rx < a.box[1] && ry > a.box[2] && ry < a.box[3];
      }
    }
  }
  if (valid) a.labels[p] = (hit || out_of_bounds(a, x, y)) ? 1.0f : 0.0f;
}

// mode 0 with a uniform cell index: the obstacle points are sorted by cell (cell_start[c] .. cell_start[c+1]) and the cell
// size is at least the robot radius, so every point closer than the radius lies in the 3 x 3 cells around the pose's
// cell.  The per-point predicate is the same fp32 arithmetic as check_points_kernel<0>: identical labels, 1/40 of the
// distance tests on the 300-disc map.
__global__ __launch_bounds__(SM_THREADS) void check_points_cells_kernel(const CheckArgs a) {
  const long long p = blockIdx.x * (long long)SM_THREADS + threadIdx.x;
  if (p >= a.n) return;
  const float x = a.poses[p * a.dim], y = a.poses[p * a.dim + 1];
  int cx = (int)floorf((x - a.cell_x0) / a.cell_size), cy = (int)floorf((y - a.cell_y0) / a.cell_size);
  cx = min(max(cx, 0), a.cells_x - 1);
  cy = min(max(cy, 0), a.cells_y - 1);
  bool hit = false;
  for (int yy = max(cy - 1, 0); yy <= min(cy + 1, a.cells_y - 1); ++yy) {
    const int c0 = yy * a.cells_x + max(cx - 1, 0), c1 = yy * a.cells_x + min(cx + 1, a.cells_x - 1);
    for (int k = a.cell_start[c0]; k < a.cell_start[c1 + 1]; ++k) {   // the row's cells are contiguous in the sorted array
      const float dx = a.obstacles[2 * k] - x, dy = a.obstacles[2 * k + 1] - y;
      hit |= closer_than(dx, dy, a.radius);
    }
  }
  a.labels[p] = (hit || out_of_bounds(a, x, y)) ? 1.0f : 0.0f;
}

// occupancy grid (onf_planner_image_map.ipynb cell 2): cell = int((x - origin - cell/2) / cell) truncated toward zero,
// evaluated in float64 like the reference's numpy (poses are float64 there; geometry scalars are Python doubles), so the
// labels of fp32 poses equal the reference's bit for bit (tests/golden/g16); outside [0, cols-1) x [0, rows-1) = collision
__global__ __launch_bounds__(SM_THREADS) void check_grid_kernel(const CheckArgs a) {
  const long long p = blockIdx.x * (long long)SM_THREADS + threadIdx.x;
  if (p >= a.n) return;
  const double x = a.poses[p * a.dim], y = a.poses[p * a.dim + 1];
  const double fx = (x - a.origin_x - a.cell / 2) / a.cell, fy = (y - a.origin_y - a.cell / 2) / a.cell;
  // numpy's float64 -> int32 cast truncates toward zero; far-away poses (|f| >= 2^31) are outside either way
  const int ix = fabs(fx) < 2.0e9 ? (int)fx : -1, iy = fabs(fy) < 2.0e9 ? (int)fy : -1;
  bool hit = true;
  if (ix >= 0 && iy >= 0 && iy < a.rows - 1 && ix < a.cols - 1) hit = a.grid[(long long)iy * a.cols + ix] > 0;
  a.labels[p] = hit ? 1.0f : 0.0f;
}

static int launch_check(const CheckArgs& a, int mode, hipStream_t st) {
  if (a.n == 0) return NFOPP_OK;
  const unsigned grid = (unsigned)((a.n + SM_THREADS - 1) / SM_THREADS);
  if (mode == 0) hipLaunchKernelGGL(check_points_kernel<0>, dim3(grid), dim3(SM_THREADS), 0, st, a);
  else if (mode == 1) hipLaunchKernelGGL(check_points_kernel<1>, dim3(grid), dim3(SM_THREADS), 0, st, a);
  else if (mode == 3) hipLaunchKernelGGL(check_points_cells_kernel, dim3(grid), dim3(SM_THREADS), 0, st, a);
  else hipLaunchKernelGGL(check_grid_kernel, dim3(grid), dim3(SM_THREADS), 0, st, a);
  NFOPP_HIP(hipGetLastError());
  return NFOPP_OK;
}

}  // namespace nfopp

using namespace nfopp;

static int fill_common(CheckArgs* a, const float* poses_dev, int64_t n, int32_t pose_dim, const float* bounds4,
                       float* labels_dev) {
  NFOPP_REQUIRE(n >= 0 && (pose_dim == 2 || pose_dim == 3), "need n >= 0 and pose_dim 2 or 3");
  NFOPP_REQUIRE(n == 0 || (poses_dev && labels_dev), "null device pointer");
  a->poses = poses_dev; a->n = n; a->dim = pose_dim; a->labels = labels_dev;
  a->has_bounds = bounds4 ? 1 : 0;
  for (int k = 0; k < 4; ++k) a->bounds[k] = bounds4 ? bounds4[k] : 0.f;
  return NFOPP_OK;
}

extern "C" int nfopp_check_collision_circle(const float* poses_dev, int64_t n, int32_t pose_dim,
                                            const float* obstacles_dev, int32_t n_obstacles, float radius,
                                            const float* bounds4, float* labels_dev, void* stream) {
  CheckArgs a = {};
  int rc = fill_common(&a, poses_dev, n, pose_dim, bounds4, labels_dev);
  if (rc) return rc;
  NFOPP_REQUIRE(n_obstacles >= 0 && (n_obstacles == 0 || obstacles_dev), "bad obstacle array");
  a.obstacles = obstacles_dev; a.n_obstacles = n_obstacles; a.radius = radius;
  return launch_check(a, 0, (hipStream_t)stream);
}

extern "C" int nfopp_check_collision_circle_cells(const float* poses_dev, int64_t n, int32_t pose_dim,
                                                  const float* obstacles_sorted_dev, int32_t n_obstacles,
                                                  const int32_t* cell_start_dev, int32_t cells_x, int32_t cells_y,
                                                  float cell_x0, float cell_y0, float cell_size, float radius,
                                                  const float* bounds4, float* labels_dev, void* stream) {
  CheckArgs a = {};
  int rc = fill_common(&a, poses_dev, n, pose_dim, bounds4, labels_dev);
  if (rc) return rc;
  NFOPP_REQUIRE(n_obstacles > 0 && obstacles_sorted_dev && cell_start_dev, "bad obstacle index");
  NFOPP_REQUIRE(cells_x > 0 && cells_y > 0 && cell_size >= radius && radius > 0.f,
                "the cell size must be at least the robot radius");
  a.obstacles = obstacles_sorted_dev; a.n_obstacles = n_obstacles; a.radius = radius;
  a.cell_start = cell_start_dev; a.cells_x = cells_x; a.cells_y = cells_y;
  a.cell_x0 = cell_x0; a.cell_y0 = cell_y0; a.cell_size = cell_size;
  return launch_check(a, 3, (hipStream_t)stream);
}

extern "C" int nfopp_check_collision_rectangle(const float* poses_dev, int64_t n, const float* obstacles_dev,
                                               int32_t n_obstacles, const float* box4, const float* bounds4,
                                               float* labels_dev, void* stream) {
  CheckArgs a = {};
  int rc = fill_common(&a, poses_dev, n, 3, bounds4, labels_dev);
  if (rc) return rc;
  NFOPP_REQUIRE(box4, "null box");
  NFOPP_REQUIRE(n_obstacles >= 0 && (n_obstacles == 0 || obstacles_dev), "bad obstacle array");
  a.obstacles = obstacles_dev; a.n_obstacles = n_obstacles;
  for (int k = 0; k < 4; ++k) a.box[k] = box4[k];
  return launch_check(a, 1, (hipStream_t)stream);
}

extern "C" int nfopp_check_collision_grid(const float* poses_dev, int64_t n, int32_t pose_dim, const uint8_t* grid_dev,
                                          int32_t rows, int32_t cols, double origin_x, double origin_y, double cell_size,
                                          float* labels_dev, void* stream) {
  CheckArgs a = {};
  int rc = fill_common(&a, poses_dev, n, pose_dim, nullptr, labels_dev);
  if (rc) return rc;
  NFOPP_REQUIRE(grid_dev && rows > 1 && cols > 1 && cell_size > 0.0, "bad occupancy grid");
  a.grid = grid_dev; a.rows = rows; a.cols = cols; a.origin_x = origin_x; a.origin_y = origin_y; a.cell = cell_size;
  return launch_check(a, 2, (hipStream_t)stream);
}

extern "C" int nfopp_sample_candidates(const float* prev_traj_dev, int64_t batch, int32_t n_waypoints, int32_t dim,
                                       int32_t pool_cap, int32_t pool_count, int32_t n_field, float course_sigma,
                                       float fine_sigma, float angle_sigma, const float* bounds4, uint64_t seed,
                                       uint64_t rng_offset, int64_t traj_index_offset, const float* pool_dev,
                                       const float* pool_age_dev, float* cand_dev, float* cand_age_dev,
                                       float* samples_dev, void* stream) {
  NFOPP_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
  NFOPP_REQUIRE(batch >= 0 && batch <= 0x7fffffffLL && n_waypoints >= 2, "bad batch / waypoint count");
  NFOPP_REQUIRE(pool_cap >= 0 && pool_cap <= n_waypoints - 1, "pool capacity must be in [0, N-1]");
  NFOPP_REQUIRE(pool_count == 0 || pool_count == pool_cap, "pool is either empty (first step) or full");
  NFOPP_REQUIRE(n_field >= 0 && bounds4, "bad field-sample arguments");
  if (batch == 0) return NFOPP_OK;
  NFOPP_REQUIRE(prev_traj_dev && cand_dev && cand_age_dev && samples_dev && (pool_count == 0 || (pool_dev && pool_age_dev)),
                "null device pointer");
  SampleArgs a = {};
  a.prev_traj = prev_traj_dev; a.batch = batch; a.traj_index_offset = traj_index_offset;
  a.n = n_waypoints; a.dim = dim; a.cap = pool_cap; a.pool_n = pool_count; a.n_field = n_field;
  a.course_sigma = course_sigma; a.fine_sigma = fine_sigma; a.angle_sigma = angle_sigma;
  for (int k = 0; k < 4; ++k) a.bounds[k] = bounds4[k];
  a.seed = seed; a.offset = rng_offset; a.pool = pool_dev; a.pool_age = pool_age_dev;
  a.cand = cand_dev; a.cand_age = cand_age_dev; a.samples = samples_dev;
  if (dim == 3) hipLaunchKernelGGL(sample_candidates_kernel<3>, dim3((unsigned)batch), dim3(SM_THREADS), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(sample_candidates_kernel<2>, dim3((unsigned)batch), dim3(SM_THREADS), 0, (hipStream_t)stream, a);
  NFOPP_HIP(hipGetLastError());
  return NFOPP_OK;
}

extern "C" int nfopp_resample_pool(int64_t batch, int32_t n_candidates, int32_t cand_stride, int32_t pool_cap,
                                   int32_t dim, int32_t sample_stride, int32_t sample_offset, uint64_t seed, uint64_t rng_offset,
                                   int64_t traj_index_offset, const float* cand_dev, const float* cand_age_dev,
                                   const float* onf_out4_dev, float* pool_dev, float* pool_age_dev, float* samples_dev,
                                   void* stream) {
  NFOPP_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
  NFOPP_REQUIRE(batch >= 0 && batch <= 0x7fffffffLL, "bad batch");
  NFOPP_REQUIRE(pool_cap >= 0 && n_candidates >= pool_cap && n_candidates <= 16384, "need pool_cap <= candidates <= 16384");
  NFOPP_REQUIRE(cand_stride >= n_candidates, "candidate stride smaller than the candidate count");
  if (batch == 0 || pool_cap == 0) return NFOPP_OK;
  NFOPP_REQUIRE(cand_dev && cand_age_dev && onf_out4_dev && pool_dev && pool_age_dev && samples_dev, "null device pointer");
  ResampleArgs a = {};
  a.batch = batch; a.traj_index_offset = traj_index_offset; a.n_cand = n_candidates; a.cand_stride = cand_stride;
  a.cap = pool_cap; a.dim = dim;
  a.sample_stride = sample_stride; a.sample_off = sample_offset; a.seed = seed; a.offset = rng_offset;
  a.cand = cand_dev; a.cand_age = cand_age_dev; a.logits = onf_out4_dev; a.pool = pool_dev; a.pool_age = pool_age_dev;
  a.samples = samples_dev;
  int n2 = 1;
  while (n2 < n_candidates) n2 <<= 1;
  const size_t lds = (size_t)n2 * 8;
  auto kern = dim == 3 ? resample_pool_kernel<3> : resample_pool_kernel<2>;
  if (lds > 64 * 1024)
    NFOPP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(SM_THREADS), lds, (hipStream_t)stream, a);
  NFOPP_HIP(hipGetLastError());
  return NFOPP_OK;
}
