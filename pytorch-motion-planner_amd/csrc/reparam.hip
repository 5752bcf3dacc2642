// Arc-length reparametrisation kernel (HBM-bound; one workgroup per trajectory; runs every
// `reparametrize_trajectory_freq` steps).
//
// Replaces nfop/constrained_nerf_opt_planner.py:132-171 (SE(2): waypoints + both multiplier arrays) and
// nfop/nerf_opt_planner.py:224-244 (2-D): xy segment lengths -> normalised cumulative distribution ->
// searchsorted(left) of the uniform grid -> linear interpolation (theta along the wrapped difference).
//
// searchsorted is INDEX work: the cdf must equal torch's bit for bit or a grid value that ties with a cdf entry lands
// on the other side of a flat run.  So the three roundings that build the cdf are torch-CPU's (each checked against
// torch in tests/test_oracle_golden.py::test_torch_reduction_orders and pinned by tests/golden/g4_reparam[clamp]):
//   * torch.norm(dim=1) of an (dx, dy) row = sqrt(fma(dy, dy, rn(dx*dx)))   (NormTwoOps `acc + data*data`, contracted)
//   * torch.sum of N+1 floats = ATen's cascade sum (SumKernel.cpp, the 8-float-vector build): 8 lane columns, four
//     interleaved accumulator chains per lane folded every 16 rows, then the scalar tail, then the lanes in order
//   * torch.cumsum accumulates in float64 (at::acc_type<float, false>) and rounds every partial sum to fp32
#include "common.h"

namespace nfopp {

constexpr int RP_THREADS = 256;

// ATen row_sum (native/cpu/SumKernel.cpp): element i = a[i * stride]; ilp_factor 4, cascade levels of 16 rows.
// level_power = max(4, ceil_log2(size / 4) / 4) = 4 for every size below 2^21 elements (LDS bounds N far below that).
__device__ __forceinline__ float torch_row_sum(const float* a, int stride, int size) {
  float acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[j][k] = 0.f;
  const int rows = size / 4;
  int i = 0;
  while (i + 16 <= rows) {
    for (int j = 0; j < 16; ++j, ++i) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[0][k] += a[(i * 4 + k) * stride];
    }
    bool more = true;
#pragma unroll
    for (int j = 1; j < 4; ++j) {
      if (more) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc[j][k] += acc[j - 1][k]; acc[j - 1][k] = 0.f; }
        if ((i & (15 << (4 * j))) != 0) more = false;
      }
    }
  }
  for (; i < rows; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[0][k] += a[(i * 4 + k) * stride];
  }
#pragma unroll
  for (int j = 1; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[0][k] += acc[j][k];
  for (int r = rows * 4; r < size; ++r) acc[0][0] += a[r * stride];
  return ((acc[0][0] + acc[0][1]) + acc[0][2]) + acc[0][3];
}

struct ReparamArgs {
  int n, dim;
  float* traj;
  const float* start;
  const float* goal;
  float* lam;
  float* cm;
  const float* u;
  const unsigned char* active;
};

template <int D>
__global__ __launch_bounds__(RP_THREADS) void reparam_kernel(const ReparamArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int N = a.n, tid = threadIdx.x;
  const long long b = blockIdx.x;
  if (a.active && !a.active[b]) return;  // retired trajectory
  float* Q = sm;                  // (N+2)*D
  float* cdf = Q + (N + 2) * D;   // N+2
  float* cmf = cdf + (N + 2);     // N+2   [0, cm, 0]
  float* lf = cmf + (N + 2);      // N+2   [l0, mid-averages, lN]
  float* li = lf + (N + 2);       // N     interpolated multipliers
  float* red = li + N;            // 8 lane sums of the torch-order reduction, the total, 4 wave sums (float64)

  float* traj = a.traj + b * N * D;
  for (int k = tid; k < N * D; k += RP_THREADS) Q[D + k] = traj[k];
  if (tid < D) {
    Q[tid] = a.start[b * D + tid];
    Q[(N + 1) * D + tid] = a.goal[b * D + tid];
  }
  if (D == 3) {
    const float* lam = a.lam + b * (N + 1);
    const float* cm = a.cm + b * N;
    for (int k = tid; k < N + 2; k += RP_THREADS) {
      cmf[k] = (k == 0 || k == N + 1) ? 0.0f : cm[k - 1];
      lf[k] = k == 0 ? lam[0] : (k == N + 1 ? lam[N] : (lam[k - 1] + lam[k]) / 2.0f);
    }
  }
  __syncthreads();

  // segment lengths (xy only, constrained:45-47), torch.norm rounding
  for (int s = tid; s <= N; s += RP_THREADS) {
    const float dx = Q[(s + 1) * D] - Q[s * D], dy = Q[(s + 1) * D + 1] - Q[s * D + 1];
    cdf[s + 1] = sqrtf(__builtin_fmaf(dy, dy, dx * dx));
  }
  __syncthreads();
  // torch.sum(distances): vectorized_inner_sum with 8-float vectors when there are at least 8 elements
  const int n_el = N + 1, n_vec = n_el >= 8 ? n_el / 8 : 0;
  if (tid < 8 && n_vec > 0) red[tid] = torch_row_sum(cdf + 1 + tid, 8, n_vec);
  __syncthreads();
  if (tid == 0) {
    float total;
    if (n_vec > 0) {
      total = 0.f;
      for (int k = n_vec * 8; k < n_el; ++k) total += cdf[1 + k];
      for (int l = 0; l < 8; ++l) total += red[l];
    } else {
      total = torch_row_sum(cdf + 1, 1, n_el);   // scalar_inner_sum
    }
    red[8] = total;
    cdf[0] = 0.f;
  }
  __syncthreads();
  // torch.cumsum on CPU: a float64 accumulator walks the fp32 quotients in order and every partial sum is rounded to fp32.
  // The quotients are multiples of 2^(e-23) with e their smallest exponent and every partial sum stays below 2, so if the
  // smallest non-zero quotient is at least 2^-29 EVERY sum of a subset of them is exactly representable in float64: the
  // additions are exact, their order does not matter, and a parallel scan returns the sequential loop's partial sums bit
  // for bit.  Otherwise (a segment 2^-29 of the path length, or a degenerate path) one lane walks the sequence as torch does.
  const float total = red[8];
  const int per = (n_el + RP_THREADS - 1) / RP_THREADS, lo_s = 1 + tid * per, hi_s = min(lo_s + per, n_el + 1);
  float qmin = 1.0f;
  bool ok = total > 0.0f && total < 3.0e38f;
  for (int s2 = lo_s; s2 < hi_s; ++s2) {
    const float q = cdf[s2] / total;
    cdf[s2] = q;
    if (q != 0.0f) qmin = fminf(qmin, q);
    ok = ok && (q >= 0.0f) && (q <= 1.0f);    // (false for NaN)
  }
  ok = ok && qmin >= 1.862645149230957e-09f;   // 2^-29
  const bool exact = __syncthreads_and(ok);
  if (exact) {
    double part = 0.0;
    for (int s2 = lo_s; s2 < hi_s; ++s2) part += (double)cdf[s2];
    // exclusive offsets of the per-thread sums: wave scan, then the wave totals through LDS (as doubles in `red2`)
    double incl = part;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double up = __shfl_up(incl, o);
      if (lane >= o) incl += up;
    }
    float* r8 = red + 9;
    if (reinterpret_cast<size_t>(r8) & 7) r8 += 1;        // 8-byte aligned slot for the four wave sums
    double* red2 = reinterpret_cast<double*>(r8);
    if (lane == 63) red2[wave] = incl;
    __syncthreads();
    double off = incl - part;
    for (int w2 = 0; w2 < wave; ++w2) off += red2[w2];
    double run = off;
    for (int s2 = lo_s; s2 < hi_s; ++s2) {
      run += (double)cdf[s2];
      cdf[s2] = (float)run;
    }
  } else if (tid == 0) {
    double run = 0.0;
    for (int s2 = 1; s2 <= N + 1; ++s2) {
      run += (double)cdf[s2];
      cdf[s2] = (float)run;
    }
  }
  __syncthreads();

  for (int w = tid; w < N; w += RP_THREADS) {
    const float u = a.u[w];
    int lo = 0, hi = N + 2;  // first index with cdf[idx] >= u (torch.searchsorted, right=False)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] < u) lo = mid + 1; else hi = mid;
    }
    const int ia = lo > N + 1 ? N + 1 : lo;
    const int ib = lo - 1 < 0 ? 0 : lo - 1;
    const float cb = cdf[ib];
    float den = cdf[ia] - cb;
    if (den < 1e-5f) den = 1e-5f;
    const float tau = (u - cb) / den;
    const float omt = 1.0f - tau;
    // products and sums rounded one by one, as the reference's separate torch ops are (no contraction to fma)
    traj[w * D] = mix_unfused(omt, Q[ib * D], tau, Q[ia * D]);
    traj[w * D + 1] = mix_unfused(omt, Q[ib * D + 1], tau, Q[ia * D + 1]);
    if (D == 3) {
      const float thb = Q[ib * 3 + 2];
      traj[w * 3 + 2] = add_mul_unfused(thb, tau, wrap_angle(Q[ia * 3 + 2] - thb));
      a.cm[b * N + w] = mix_unfused(omt, cmf[ib], tau, cmf[ia]);
      li[w] = mix_unfused(omt, lf[ib], tau, lf[ia]);
    }
  }
  if (D == 3) {
    __syncthreads();
    float* lam = a.lam + b * (N + 1);
    for (int k = tid; k <= N; k += RP_THREADS)
      lam[k] = k == 0 ? li[0] : (k == N ? li[N - 1] : (li[k - 1] + li[k]) / 2.0f);
  }
}

}  // namespace nfopp

using namespace nfopp;

extern "C" int nfopp_reparametrize(int64_t batch, int32_t n_waypoints, int32_t dim, float* traj_dev,
                                   const float* start_dev, const float* goal_dev, float* lam_dev, float* cm_dev,
                                   const float* u_dev, const uint8_t* active_dev, void* stream) {
  NFOPP_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
  NFOPP_REQUIRE(batch >= 0 && n_waypoints >= 2, "need batch >= 0 and at least 2 waypoints");
  NFOPP_REQUIRE(batch <= 0x7fffffffLL, "batch too large for one launch");
  if (batch == 0) return NFOPP_OK;
  NFOPP_REQUIRE(traj_dev && start_dev && goal_dev && u_dev, "null device pointer");
  NFOPP_REQUIRE(dim == 2 || (lam_dev && cm_dev), "SE(2) reparametrisation needs the multiplier arrays");
  ReparamArgs a;
  a.n = n_waypoints; a.dim = dim; a.traj = traj_dev; a.start = start_dev; a.goal = goal_dev;
  a.lam = lam_dev; a.cm = cm_dev; a.u = u_dev; a.active = active_dev;
  const size_t lds = (size_t)((n_waypoints + 2) * dim + 3 * (n_waypoints + 2) + n_waypoints + 8 + 12) * 4;
  NFOPP_REQUIRE(lds <= 160 * 1024, "trajectory too long for one workgroup's LDS (%zu bytes)", lds);
  auto kern = dim == 3 ? reparam_kernel<3> : reparam_kernel<2>;
  if (lds > 64 * 1024)
    NFOPP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(RP_THREADS), lds, (hipStream_t)stream, a);
  NFOPP_HIP(hipGetLastError());
  return NFOPP_OK;
}
