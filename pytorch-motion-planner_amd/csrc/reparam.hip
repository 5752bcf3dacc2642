// Arc-length reparametrisation kernel (HBM-bound; one workgroup per trajectory; runs every
// `reparametrize_trajectory_freq` steps).
//
// Replaces nfop/constrained_nerf_opt_planner.py:132-171 (SE(2): waypoints + both multiplier arrays) and
// nfop/nerf_opt_planner.py:224-244 (2-D): xy segment lengths -> normalised cumulative distribution ->
// searchsorted(left) of the uniform grid -> linear interpolation (theta along the wrapped difference).
// The cumulative sum is accumulated sequentially in fp32 like torch.cumsum on CPU.
#include "common.h"

namespace nfopp {

constexpr int RP_THREADS = 256;

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
  float* red = li + N;            // RP_THREADS/64

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

  // segment lengths (xy only, constrained:45-47) and their sum
  float part = 0.f;
  for (int s = tid; s <= N; s += RP_THREADS) {
    const float dx = Q[(s + 1) * D] - Q[s * D], dy = Q[(s + 1) * D + 1] - Q[s * D + 1];
    const float d = sqrtf(dx * dx + dy * dy);
    cdf[s + 1] = d;
    part += d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
  if ((tid & 63) == 0) red[tid >> 6] = part;
  __syncthreads();
  float total = 0.f;
  for (int w = 0; w < RP_THREADS / 64; ++w) total += red[w];
  if (tid == 0) {
    float run = 0.f;
    cdf[0] = 0.f;
    for (int s = 1; s <= N + 1; ++s) {
      run += cdf[s] / total;
      cdf[s] = run;
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
    traj[w * D] = omt * Q[ib * D] + tau * Q[ia * D];
    traj[w * D + 1] = omt * Q[ib * D + 1] + tau * Q[ia * D + 1];
    if (D == 3) {
      const float thb = Q[ib * 3 + 2];
      traj[w * 3 + 2] = thb + tau * wrap_angle(Q[ia * 3 + 2] - thb);
      a.cm[b * N + w] = omt * cmf[ib] + tau * cmf[ia];
      li[w] = omt * lf[ib] + tau * lf[ia];
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
  const size_t lds = (size_t)((n_waypoints + 2) * dim + 3 * (n_waypoints + 2) + n_waypoints + RP_THREADS / 64) * 4;
  NFOPP_REQUIRE(lds <= 160 * 1024, "trajectory too long for one workgroup's LDS (%zu bytes)", lds);
  auto kern = dim == 3 ? reparam_kernel<3> : reparam_kernel<2>;
  if (lds > 64 * 1024)
    NFOPP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(RP_THREADS), lds, (hipStream_t)stream, a);
  NFOPP_HIP(hipGetLastError());
  return NFOPP_OK;
}
