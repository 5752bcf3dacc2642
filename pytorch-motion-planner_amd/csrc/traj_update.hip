// Per-trajectory stencil + preconditioner + optimiser kernel (HBM-bound; one workgroup per trajectory).
//
// Replaces (reference, PyTorch-CPU + autograd): the non-ONF part of `trajectory_loss` and its backward
// nfop/constrained_nerf_opt_planner.py:82-130, nfop/nerf_opt_planner.py:171-176 (SE(2)) or
// nfop/nerf_opt_planner.py:157-169 (2-D); `grad = inv_hessian @ grad` nerf:151; torch.optim.Adam nerf:154;
// multiplier ascent + clamp constrained:66-73.  Closed-form gradients: SURVEY.md Appendix A.
//
// Every waypoint's gradient is assembled in registers from its two adjacent segments (3-point stencil) and its
// two adjacent collision samples, parked once in LDS for the banded H^-1 product, and the state
// (traj, m, v, lambda, cm) is read once and written once: 92 B + 16 B (ONF record) per waypoint-step.
#include "common.h"

namespace nfopp {

constexpr int TU_THREADS = 256;
// LDS budget for the band columns of the boundary waypoints.  Their number and the band width both grow with the
// velocity-Hessian weight (a user hyper-parameter the reference accepts at any value: w = 10, N = 512 needs 205 KB);
// beyond the budget those waypoints read their coefficients from global memory tap by tap.
constexpr size_t K2_BND_BUDGET = 32 * 1024;
#ifndef NFOPP_K2_WAVES
#define NFOPP_K2_WAVES 1   /* minimum waves per SIMD the register allocation aims at (A/B: 6 and 8 below) */
#endif

struct TrajUpdateArgs {
  nfopp_traj_hyper hp;
  long long batch;
  int n, dim;
  float* traj;
  const float* start;
  const float* goal;
  float* lam;
  float* cm;
  float* adam_m;
  float* adam_v;
  const float* t;
  const float* onf;
  const float* hinv_band;
  int half_width;
  int interior_lo, interior_hi;  // waypoints whose band column equals column interior_lo bit for bit (Toeplitz interior)
  int bnd_in_lds;                // the boundary waypoints' band columns are staged in LDS (they fit K2_BND_BUDGET)
  float* terms;
  const unsigned char* active;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// sums K per-thread values over the workgroup (all threads get the result); scratch >= K * (threads / 64) floats
template <int K>
__device__ void block_sum(float (&v)[K], float* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float s = wave_sum(v[k]);
    if (lane == 0) scratch[wave * K + k] = s;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float s = 0.f;
    for (int w = 0; w < nw; ++w) s += scratch[w * K + k];
    v[k] = s;
  }
}

struct SegOut {  // contribution of one segment to its head (q_{s+1}) and tail (q_s) waypoint, plus its loss terms
  float hx, hy, hth, tx, ty, tth;
  float c, l_dist, l_dir;
};

// segment s joins q_s=(x0,y0,th0) -> q_{s+1}=(x1,y1,th1); `extra` = winding correction (last segment only)
__device__ __forceinline__ SegOut segment_terms(const nfopp_traj_hyper& hp, float x0, float y0, float th0, float x1,
                                                float y1, float th1, float lam_s, float extra) {
  SegOut o;
  const float dx = x1 - x0, dy = y1 - y0;
  // A7 distance (constrained:120-130): raw theta difference (+ winding constant), scaled by angle_weight
  const float aw = hp.angle_weight;
  const float dthw = ((th1 - th0) + extra) * aw;
  o.l_dist = dx * dx + dy * dy + dthw * dthw;
  o.hx = 2.0f * dx; o.hy = 2.0f * dy; o.hth = (2.0f * dthw) * aw;
  o.tx = -o.hx; o.ty = -o.hy; o.tth = -o.hth;
  // A5 non-holonomic (constrained:102-109)
  const float m = th0 + wrap_angle(th1 - th0) * 0.5f;
  const float sm = sin_halfturns_hw(m, 0.0f), cm_ = sin_halfturns_hw(m, 0.25f);
  const float c = dx * sm - dy * cm_;
  const float e = dx * cm_ + dy * sm;
  const float gg = lam_s + (2.0f * hp.constraint_deltas_weight) * c;
  o.c = c;
  o.hx += gg * sm; o.tx -= gg * sm;
  o.hy -= gg * cm_; o.ty += gg * cm_;
  o.hth += gg * e * 0.5f; o.tth += gg * e * 0.5f;
  // A6 direction / forward-only (constrained:111-118,93,98)
  const float mp = th0 + wrap_angle(th0 - th1) * 0.5f;
  const float smp = sin_halfturns_hw(mp, 0.0f), cmp_ = sin_halfturns_hw(mp, 0.25f);
  const float d = -(cmp_ * dx + smp * dy);
  const float r = d > 0.0f ? d : 0.0f;
  o.l_dir = r * r;
  const float hh = (2.0f * hp.direction_delta_weight) * r;
  const float k = smp * dx - cmp_ * dy;
  o.hx -= hh * cmp_; o.tx += hh * cmp_;
  o.hy -= hh * smp; o.ty += hh * smp;
  o.tth += 1.5f * hh * k; o.hth -= 0.5f * hh * k;
  return o;
}

// collision sample j of this trajectory: returns gamma (upstream on the logit), tanh(logit), softplus value
__device__ __forceinline__ void collision_terms(const nfopp_traj_hyper& hp, float logit, float cm_i, float* gamma,
                                                float* th_l, float* sp) {
  const float beta = hp.collision_beta;
  const float bl = logit * beta;
  const float z = expf(bl);
  const bool lin = bl > 20.0f;  // torch softplus threshold
  *sp = lin ? logit : log1pf(z) / beta;
  const float dsp = lin ? 1.0f : z / (z + 1.0f);
  const float th = tanhf(logit);
  *th_l = th;
  *gamma = hp.collision_weight * dsp + cm_i * (1.0f - th * th);
}

template <int D>
__global__ __launch_bounds__(TU_THREADS, NFOPP_K2_WAVES) void traj_update_kernel(const TrajUpdateArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int N = a.n;
  const int W = a.half_width, NB_LO = a.interior_lo;
  const int NB = a.bnd_in_lds ? a.interior_lo + (N - a.interior_hi) : 0;  // boundary columns staged in LDS
  float* Q = sm;                    // (N+2) * D full trajectory
  float* GP = Q + (N + 2) * D;      // (N + 2W) * D gradient, W zero rows on either side: every tap of every waypoint is
  float* G = GP + W * D;            //   in range, so the band loop is uniform over the workgroup
  float* LAM = GP + (N + 2 * W) * D;  // N+1 constraint multipliers (read-only copy: the update writes global)
  float* CM = LAM + (N + 1);        // N collision multipliers
  float* COEF = CM + N;             // 2W+1 interior band coefficients (Toeplitz interior: one column for all)
  float* BND = COEF + (2 * W + 1);  // NB * (2W+1): the band columns of the waypoints near the two ends
  float* scratch = BND + NB * (2 * W + 1);  // block reductions
  const nfopp_traj_hyper& hp = a.hp;
  const int tid = threadIdx.x;

  // once per workgroup: preconditioner coefficients (the boundary waves used to fetch their 2W+1 coefficients from global
  // memory inside the band loop, 4 in flight: 27 % of the kernel) and the zero rows around the gradient
  if (a.interior_hi > a.interior_lo)
    for (int k = tid; k <= 2 * W; k += TU_THREADS) COEF[k] = a.hinv_band[(long long)k * N + a.interior_lo];
  for (int idx = tid; idx < NB * (2 * W + 1); idx += TU_THREADS) {
    const int k = idx / NB, c = idx - k * NB;                 // consecutive threads: consecutive columns of one band row
    const int w = c < NB_LO ? c : a.interior_hi + (c - NB_LO);
    BND[c * (2 * W + 1) + k] = a.hinv_band[(long long)k * N + w];
  }
  for (int k = tid; k < W * D; k += TU_THREADS) { GP[k] = 0.0f; GP[(N + W) * D + k] = 0.0f; }

  // one trajectory per workgroup (walking several in a row measured slower: nothing overlaps between a workgroup's
  // trajectories, and the loop costs 36 VGPRs = one wave per SIMD)
  const long long b = blockIdx.x;
  if (a.active && !a.active[b]) return;  // retired trajectory (uniform per workgroup)

  float* traj = a.traj + b * N * D;
  // This thread's first waypoint (w = tid) needs two draws, two ONF records and its Adam moments from global memory.  Issued
  // here, with the state loads, they cost ONE exposure to the memory latency instead of one per phase (the loads cannot
  // move across the barriers by themselves).  Waypoints beyond the first pass (N > 256) load where they are used.
  const float* tb = a.t + b * (N - 1);
  const float* onf = a.onf + b * (N - 1) * 4;
  float* am = a.adam_m + b * N * D;
  float* av = a.adam_v + b * N * D;
  float pf_t0 = 0.f, pf_t1 = 0.f, pf_m[D], pf_v[D];
  float4 pf_o0 = {0.f, 0.f, 0.f, 0.f}, pf_o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int d = 0; d < D; ++d) { pf_m[d] = 0.f; pf_v[d] = 0.f; }
  if (tid < N) {
    if (tid <= N - 2) { pf_t0 = tb[tid]; pf_o0 = reinterpret_cast<const float4*>(onf)[tid]; }
    if (tid >= 1) { pf_t1 = tb[tid - 1]; pf_o1 = reinterpret_cast<const float4*>(onf)[tid - 1]; }
#pragma unroll
    for (int d = 0; d < D; ++d) { pf_m[d] = am[tid * D + d]; pf_v[d] = av[tid * D + d]; }
  }
  for (int k = tid; k < N * D; k += TU_THREADS) Q[D + k] = traj[k];
  if (tid < D) {
    Q[tid] = a.start[b * D + tid];
    Q[(N + 1) * D + tid] = a.goal[b * D + tid];
  }
  if (D == 3) {
    for (int k = tid; k <= N; k += TU_THREADS) LAM[k] = a.lam[b * (N + 1) + k];
    for (int k = tid; k < N; k += TU_THREADS) CM[k] = a.cm[b * N + k];
  }
  __syncthreads();

  // (staging t / the ONF records in LDS with the state loads measured slower, twice: round 1 and round 2; round 3 keeps
  // them in registers instead, see the prefetch above)
  float terms[NFOPP_NUM_TERMS];
#pragma unroll
  for (int k = 0; k < NFOPP_NUM_TERMS; ++k) terms[k] = 0.f;

  if (D == 3) {
    // winding constant C = sum_s wrap(dtheta_s) - theta_goal + theta_start (constrained:124-125)
    float part[1] = {0.f};
    for (int s = tid; s <= N; s += TU_THREADS) part[0] += wrap_angle(Q[(s + 1) * 3 + 2] - Q[s * 3 + 2]);
    block_sum<1>(part, scratch);
    const float C = part[0] - Q[(N + 1) * 3 + 2] + Q[2];
    const float* lam = LAM;
    const float* cm = CM;

    for (int w = tid; w < N; w += TU_THREADS) {
      const int f = w + 1;
      const float xp = Q[(f - 1) * 3], yp = Q[(f - 1) * 3 + 1], tp = Q[(f - 1) * 3 + 2];
      const float xc = Q[f * 3], yc = Q[f * 3 + 1], tc = Q[f * 3 + 2];
      const float xn = Q[(f + 1) * 3], yn = Q[(f + 1) * 3 + 1], tn = Q[(f + 1) * 3 + 2];
      const float lam_l = lam[w], lam_r = lam[w + 1];
      const SegOut L = segment_terms(hp, xp, yp, tp, xc, yc, tc, lam_l, 0.0f);                    // s = w, we are head
      const SegOut R = segment_terms(hp, xc, yc, tc, xn, yn, tn, lam_r, (w == N - 1) ? C : 0.0f);  // s = w+1, tail
      float gx = L.hx + R.tx, gy = L.hy + R.ty, gth = L.hth + R.tth;
      terms[1] += L.l_dist; terms[3] += lam_l * L.c; terms[4] += L.c * L.c; terms[7] += L.l_dir;
      if (w == N - 1) { terms[1] += R.l_dist; terms[3] += lam_r * R.c; terms[4] += R.c * R.c; terms[7] += R.l_dir; }
      // A8 boundary (nerf:171-176)
      const float bx0 = fmaxf(hp.bounds[0] - xc, 0.f), bx1 = fmaxf(xc - hp.bounds[1], 0.f);
      const float by0 = fmaxf(hp.bounds[2] - yc, 0.f), by1 = fmaxf(yc - hp.bounds[3], 0.f);
      terms[5] += bx0 * bx0 + bx1 * bx1 + by0 * by0 + by1 * by1;
      const float wb = 2.0f * hp.boundary_weight;
      gx += wb * (bx1 - bx0);
      gy += wb * (by1 - by0);
      // A4 collision samples j = w (we are `a`, weight t) and j = w-1 (we are `b`, weight 1-t)
      float g_cm = 0.f;
      const float cm_c = cm[w];
      if (w <= N - 2) {
        const bool first = w == tid;
        const float tj = first ? pf_t0 : tb[w];
        const float4 o = first ? pf_o0 : reinterpret_cast<const float4*>(onf)[w];
        const float cm_i = cm[w + 1] * (1.0f - tj) + cm_c * tj;
        float gamma, thl, sp;
        collision_terms(hp, o.x, cm_i, &gamma, &thl, &sp);
        terms[2] += sp; terms[6] += cm_i * thl;
        gx += tj * (gamma * o.y); gy += tj * (gamma * o.z); gth += tj * (gamma * o.w);
        g_cm += tj * thl;
      }
      if (w >= 1) {
        const bool first = w == tid;
        const float tj = first ? pf_t1 : tb[w - 1];
        const float4 o = first ? pf_o1 : reinterpret_cast<const float4*>(onf)[w - 1];
        const float cm_i = cm_c * (1.0f - tj) + cm[w - 1] * tj;
        float gamma, thl, sp;
        collision_terms(hp, o.x, cm_i, &gamma, &thl, &sp);
        const float omt = 1.0f - tj;
        gx += omt * (gamma * o.y); gy += omt * (gamma * o.z); gth += omt * (gamma * o.w);
        g_cm += omt * thl;
      }
      G[w * 3] = gx; G[w * 3 + 1] = gy; G[w * 3 + 2] = gth;
      // multiplier ascent (constrained:66-73); lambda_N belongs to the last thread
      a.lam[b * (N + 1) + w] = lam_l + hp.multipliers_lr * L.c;
      if (w == N - 1) a.lam[b * (N + 1) + N] = lam_r + hp.multipliers_lr * R.c;
      const float cm_new = cm_c + hp.collision_multipliers_lr * g_cm;
      a.cm[b * N + w] = cm_new > 0.0f ? cm_new : 0.0f;
    }
  } else {
    // 2-D planner: squared segment lengths + collision_weight * softplus(logit) (nerf:157-169)
    for (int w = tid; w < N; w += TU_THREADS) {
      const int f = w + 1;
      const float xp = Q[(f - 1) * 2], yp = Q[(f - 1) * 2 + 1];
      const float xc = Q[f * 2], yc = Q[f * 2 + 1];
      const float xn = Q[(f + 1) * 2], yn = Q[(f + 1) * 2 + 1];
      const float dlx = xc - xp, dly = yc - yp, drx = xn - xc, dry = yn - yc;
      float gx = 2.0f * dlx - 2.0f * drx, gy = 2.0f * dly - 2.0f * dry;
      terms[1] += dlx * dlx + dly * dly;
      if (w == N - 1) terms[1] += drx * drx + dry * dry;
      if (w <= N - 2) {
        const bool first = w == tid;
        const float tj = first ? pf_t0 : tb[w];
        const float4 o = first ? pf_o0 : reinterpret_cast<const float4*>(onf)[w];
        float gamma, thl, sp;
        collision_terms(hp, o.x, 0.0f, &gamma, &thl, &sp);
        terms[2] += sp;
        gx += tj * (gamma * o.y); gy += tj * (gamma * o.z);
      }
      if (w >= 1) {
        const bool first = w == tid;
        const float tj = first ? pf_t1 : tb[w - 1];
        const float4 o = first ? pf_o1 : reinterpret_cast<const float4*>(onf)[w - 1];
        float gamma, thl, sp;
        collision_terms(hp, o.x, 0.0f, &gamma, &thl, &sp);
        const float omt = 1.0f - tj;
        gx += omt * (gamma * o.y); gy += omt * (gamma * o.z);
      }
      G[w * 2] = gx; G[w * 2 + 1] = gy;
    }
  }
  __syncthreads();

  // g <- H^-1 g with the band of the reference's fp32 inverse (nerf:151), then Adam (torch single-tensor path).
  // Taps k = 0 .. 2W in ascending order for every waypoint; out-of-range taps meet a zero band entry and a zero-padded
  // gradient row, so the sum is the same fp32 chain as over the valid taps alone.
  for (int w = tid; w < N; w += TU_THREADS) {
    float acc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = 0.f;
    // optimiser state is independent of the band product: issue its loads before the taps
    float m_in[D], v_in[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { m_in[d] = w == tid ? pf_m[d] : am[w * D + d]; v_in[d] = w == tid ? pf_v[d] : av[w * D + d]; }
    const bool interior = w >= a.interior_lo && w < a.interior_hi;
    const float* gj = GP + w * D;   // row w - W of the unpadded gradient
    if (interior || a.bnd_in_lds) {
      const float* cf = interior ? COEF : BND + (w < a.interior_lo ? w : NB_LO + (w - a.interior_hi)) * (2 * W + 1);
#pragma unroll 4
      for (int k = 0; k <= 2 * W; ++k) {
        const float hv = cf[k];
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = fmaf(hv, gj[k * D + d], acc[d]);
      }
    } else {   // boundary waypoint, columns not staged: the same taps in the same order, coefficients from global memory
      const float* cf = a.hinv_band + w;
#pragma unroll 4
      for (int k = 0; k <= 2 * W; ++k) {
        const float hv = cf[(long long)k * N];
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = fmaf(hv, gj[k * D + d], acc[d]);
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const float g = acc[d];
      float m = m_in[d], v = v_in[d];
      m = m + hp.adam_omb1 * (g - m);
      v = v * hp.adam_beta2 + (hp.adam_omb2 * g) * g;
      const float denom = sqrtf(v) / hp.adam_bc2_sqrt + hp.adam_eps;
      const float p = Q[(w + 1) * D + d] - hp.adam_step_size * (m / denom);
      am[w * D + d] = m;
      av[w * D + d] = v;
      traj[w * D + d] = p;
    }
  }

  if (a.terms) {
    block_sum<NFOPP_NUM_TERMS>(terms, scratch);
    if (tid == 0) {
      float* o = a.terms + b * NFOPP_NUM_TERMS;
      const float total = terms[1] + hp.collision_weight * terms[2] + terms[3] +
                          hp.constraint_deltas_weight * terms[4] + hp.boundary_weight * terms[5] + terms[6] +
                          hp.direction_delta_weight * terms[7];
      o[0] = total;
#pragma unroll
      for (int k = 1; k < NFOPP_NUM_TERMS; ++k) o[k] = terms[k];
    }
  }
}

}  // namespace nfopp

using namespace nfopp;

extern "C" int nfopp_traj_update(const nfopp_traj_hyper* hp, int64_t batch, int32_t n_waypoints, int32_t dim,
                                 float* traj_dev, const float* start_dev, const float* goal_dev, float* lam_dev,
                                 float* cm_dev, float* adam_m_dev, float* adam_v_dev, const float* t_dev,
                                 const float* onf_out4_dev, const float* hinv_band_dev, int32_t half_width,
                                 int32_t interior_lo, int32_t interior_hi, float* terms_dev,
                                 const uint8_t* active_dev, void* stream) {
  NFOPP_REQUIRE(hp, "null hyper-parameter block");
  NFOPP_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
  NFOPP_REQUIRE(batch >= 0 && n_waypoints >= 2, "need batch >= 0 and at least 2 waypoints");
  NFOPP_REQUIRE(half_width >= 0, "negative band half-width");
  NFOPP_REQUIRE(interior_lo >= 0 && interior_hi <= n_waypoints, "interior range outside the trajectory");
  if (batch == 0) return NFOPP_OK;
  NFOPP_REQUIRE(traj_dev && start_dev && goal_dev && adam_m_dev && adam_v_dev && t_dev && onf_out4_dev &&
                    hinv_band_dev,
                "null device pointer");
  NFOPP_REQUIRE(dim == 2 || (lam_dev && cm_dev), "SE(2) update needs the multiplier arrays");
  TrajUpdateArgs a;
  a.hp = *hp;
  a.batch = batch; a.n = n_waypoints; a.dim = dim;
  a.traj = traj_dev; a.start = start_dev; a.goal = goal_dev; a.lam = lam_dev; a.cm = cm_dev;
  a.adam_m = adam_m_dev; a.adam_v = adam_v_dev; a.t = t_dev; a.onf = onf_out4_dev;
  a.hinv_band = hinv_band_dev; a.half_width = half_width; a.terms = terms_dev; a.active = active_dev;
  a.interior_lo = interior_lo; a.interior_hi = interior_hi > interior_lo ? interior_hi : interior_lo;
  int n_boundary = a.interior_lo + (n_waypoints - a.interior_hi);
  a.bnd_in_lds = (size_t)n_boundary * (2 * half_width + 1) * 4 <= K2_BND_BUDGET;
  if (!a.bnd_in_lds) n_boundary = 0;
  const size_t lds = (size_t)((n_waypoints + 2) * dim + (n_waypoints + 2 * half_width) * dim + 2 * n_waypoints + 1 +
                              (n_boundary + 1) * (2 * half_width + 1) + NFOPP_NUM_TERMS * (TU_THREADS / 64)) * 4;
  NFOPP_REQUIRE(lds <= 160 * 1024, "trajectory too long for one workgroup's LDS (%zu bytes)", lds);
  NFOPP_REQUIRE(batch <= 0x7fffffffLL, "batch too large for one launch");
  auto kern = dim == 3 ? traj_update_kernel<3> : traj_update_kernel<2>;
  if (lds > 64 * 1024)
    NFOPP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
  const long long grid = batch;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(TU_THREADS), lds, (hipStream_t)stream, a);
  NFOPP_HIP(hipGetLastError());
  return NFOPP_OK;
}
