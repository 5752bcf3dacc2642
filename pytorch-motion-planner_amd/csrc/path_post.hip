// Path post-processing for the follower, one workgroup per path.
//
// Replaces PathPostprocessor.process (nfop/ros/path_postprocessor.py:13-69): drop interior poses closer than
// `minimal_distance` to the previously kept one (walking from the goal), re-sample the path every `distance_step`
// metres along a QUADRATIC interpolating spline over the normalised chord-length parameter (scipy
// interp1d(kind="quadratic") = make_interp_spline(k=2): knots at the data-site midpoints, end knots tripled), headings
// unfolded first, and trim the leading poses that are driven in the opposite direction to the first one.
//
// Precision follows numpy on the fp32 path the planner returns: filter, segment lengths, their running sum and the
// unfolded headings in fp32; parametrisation, spline and output in float64.  With midpoint knots row j of the
// collocation matrix touches only coefficients j-1..j+1, so the system is tridiagonal and is eliminated without
// pivoting by one lane (scipy calls LAPACK gbsv; agreement to rounding, checked against scipy's output in
// tests/golden/g12).  The sequential parts (filter, sums, elimination: O(n)) run on lane 0; evaluation of the
// `count` output poses is spread over the workgroup.
#include "common.h"

// numpy evaluates every fp32 / float64 operation separately: no fused multiply-adds anywhere in this file
// (hipcc's `__fmul_rn` / `__fadd_rn` are plain operators and would still be contracted)
#pragma clang fp contract(off)

namespace nfopp {

constexpr int PP_THREADS = 256;

struct PostArgs {
  const float* path;   // [B, n, 3]
  int n, max_out;
  float min_dist, dist_step;
  double* out;         // [B, max_out, 3]
  int* count;          // [B]: poses the path needs (after trimming); -1 = fewer than 3 poses survive the filter
};

// numpy's pairwise float summation order (8 partial sums per block of <= 128, halves split at a multiple of 8);
// the recursion is unrolled at compile time: 4 levels cover the 1025 segments of the longest path
template <int LEVELS>
__device__ __forceinline__ float pairwise_sum(const float* a, int n) {
  if (n < 8) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s = s + a[i];
    return s;
  }
  if constexpr (LEVELS > 0) {
    if (n > 128) {
      int n2 = n / 2;
      n2 -= n2 % 8;
      return pairwise_sum<LEVELS - 1>(a, n2) + pairwise_sum<LEVELS - 1>(a + n2, n - n2);
    }
  }
  float r[8];
  for (int j = 0; j < 8; ++j) r[j] = a[j];
  int i = 8;
  for (; i < n - (n % 8); i += 8)
    for (int j = 0; j < 8; ++j) r[j] = r[j] + a[i + j];
  float s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) s = s + a[i];
  return s;
}

// np.remainder(x, 2 pi) for fp32 (result takes the sign of the divisor)
__device__ __forceinline__ float remainder_two_pi(float x) {
  float r = fmodf(x, NFOPP_TWO_PI_F);
  if (r != 0.f && r < 0.f) r = (r + NFOPP_TWO_PI_F);
  return r;
}

// the three quadratic B-spline basis values B_{ell-2..ell}(x) on knots t, t[ell] <= x < t[ell+1] (de Boor-Cox)
__device__ __forceinline__ void basis2(const double* t, int ell, double x, double h[3]) {
  h[0] = 1.0; h[1] = 0.0; h[2] = 0.0;
#pragma unroll
  for (int j = 1; j <= 2; ++j) {
    double hh[2] = {h[0], h[1]};
    h[0] = 0.0;
#pragma unroll
    for (int n = 1; n <= j; ++n) {
      const double xb = t[ell + n], xa = t[ell + n - j];
      if (xb == xa) { h[n] = 0.0; continue; }
      const double w = hh[n - 1] / (xb - xa);
      h[n - 1] += w * (xb - x);
      h[n] = w * (x - xa);
    }
  }
}

__device__ __forceinline__ void spline_at(const double* t, const double* c, int m, double x, double p[3]) {
  int lo = 2, hi = m - 1;   // largest ell in [2, m-1] with t[ell] <= x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (t[mid] <= x) lo = mid; else hi = mid - 1;
  }
  double h[3];
  basis2(t, lo, x, h);
#pragma unroll
  for (int d = 0; d < 3; ++d) p[d] = h[0] * c[(lo - 2) * 3 + d] + h[1] * c[(lo - 1) * 3 + d] + h[2] * c[lo * 3 + d];
}

__global__ __launch_bounds__(PP_THREADS) void path_post_kernel(const PostArgs a) {
  extern __shared__ double lds[];
  const int n = a.n;
  double* T = lds;              // n + 3 knots
  double* C = T + (n + 3);      // 3n spline coefficients (right-hand side in place)
  double* PAR = C + 3 * n;      // n parameter values
  double* DD = PAR + n;         // n pivots
  double* UP = DD + n;          // n super-diagonal
  float* P = reinterpret_cast<float*>(UP + n);   // 3n filtered poses
  float* DIST = P + 3 * n;      // n segment lengths
  __shared__ int s_m, s_count, s_first;
  const long long b = blockIdx.x;
  const float* src = a.path + b * (long long)n * 3;
  for (int k = threadIdx.x; k < 3 * n; k += PP_THREADS) P[k] = src[k];
  __syncthreads();

  if (threadIdx.x == 0) {
    // ---- filter (:35-44), in place from the back: kept poses end up in P[w..n-1]
    int w = n - 1;
    float px = P[3 * (n - 1)], py = P[3 * (n - 1) + 1];
    for (int i = n - 2; i >= 1; --i) {
      const float dx = px - P[3 * i], dy = py - P[3 * i + 1];
      const float dist = sqrtf(((dx * dx) + (dy * dy)));
      if (dist > a.min_dist) {
        --w;
        P[3 * w] = P[3 * i]; P[3 * w + 1] = P[3 * i + 1]; P[3 * w + 2] = P[3 * i + 2];
        px = P[3 * i]; py = P[3 * i + 1];
      }
    }
    --w;
    const float x0 = P[0], y0 = P[1], a0 = P[2];
    P[3 * w] = x0; P[3 * w + 1] = y0; P[3 * w + 2] = a0;
    const int m = n - w;
    float* Q = P + 3 * w;
    int count = -1;
    if (m >= 3) {
      // ---- parametrisation (:27-33) and total length (:21-25)
      float acc = 0.f;
      PAR[0] = 0.0;
      for (int i = 0; i < m - 1; ++i) {
        const float dx = Q[3 * (i + 1)] - Q[3 * i], dy = Q[3 * (i + 1) + 1] - Q[3 * i + 1];
        const float d = sqrtf((dx * dx) + (dy * dy)) + 1e-6f;
        DIST[i] = d;
        acc = (acc + d);
        PAR[i + 1] = (double)acc;
      }
      const double last = PAR[m - 1];
      for (int i = 0; i < m; ++i) PAR[i] = PAR[i] / last;
      const float total = pairwise_sum<4>(DIST, m - 1);
      count = (int)(total / a.dist_step);
      // ---- unfold headings (utils/math.py:38-43)
      float prev = remainder_two_pi(Q[2] + NFOPP_PI_F) - NFOPP_PI_F;
      const float first_angle = prev;
      float run = 0.f;
      Q[2] = (float)((double)first_angle);
      for (int i = 1; i < m; ++i) {
        const float cur = remainder_two_pi(Q[3 * i + 2] + NFOPP_PI_F) - NFOPP_PI_F;
        float d = (cur + -prev);
        if (d > NFOPP_PI_F) d = (d + -NFOPP_TWO_PI_F);
        if (d < -NFOPP_PI_F) d = (d + NFOPP_TWO_PI_F);
        run = (run + d);
        prev = cur;
        Q[3 * i + 2] = (float)((double)first_angle + (double)run);
      }
      // ---- knots and the tridiagonal collocation system
      T[0] = T[1] = T[2] = PAR[0];
      for (int i = 1; i <= m - 3; ++i) T[2 + i] = (PAR[i + 1] + PAR[i]) / 2.0;
      T[m] = T[m + 1] = T[m + 2] = PAR[m - 1];
      for (int i = 0; i < m; ++i)
        for (int d = 0; d < 3; ++d) C[3 * i + d] = (double)Q[3 * i + d];
      DD[0] = 1.0; UP[0] = 0.0;
      for (int j = 1; j < m; ++j) {
        double lowj = 0.0, diagj = 1.0, upj = 0.0;
        if (j < m - 1) {
          // x_j lies in knot interval ell = j+1 (clamped to [2, m-1]); its three basis functions are coefficients j-1..j+1
          const int ell = j + 1 > m - 1 ? m - 1 : j + 1;
          double h[3];
          basis2(T, ell, PAR[j], h);
          lowj = h[0]; diagj = h[1]; upj = h[2];
        }
        const double wgt = lowj / DD[j - 1];
        DD[j] = diagj - wgt * UP[j - 1];
        UP[j] = upj;
        for (int d = 0; d < 3; ++d) C[3 * j + d] -= wgt * C[3 * (j - 1) + d];
      }
      for (int d = 0; d < 3; ++d) C[3 * (m - 1) + d] /= DD[m - 1];
      for (int j = m - 2; j >= 0; --j)
        for (int d = 0; d < 3; ++d) C[3 * j + d] = (C[3 * j + d] - UP[j] * C[3 * (j + 1) + d]) / DD[j];
      // ---- direction of the first segments (:54-69): only the first 7 poses matter
      int first = 1;
      if (count >= 2) {
        const double step = 1.0 / (double)(count - 1);
        const int lim = count < 8 ? count : 8;
        double prevp[3];
        bool fwd0 = false;
        for (int q = 0; q < lim; ++q) {
          const double x = q == count - 1 ? 1.0 : (double)q * step;
          double p[3];
          spline_at(T, C, m, x, p);
          if (q > 0) {
            double dth = fmod(p[2] - prevp[2] + NFOPP_PI_D, 2.0 * NFOPP_PI_D);
            if (dth != 0.0 && dth < 0.0) dth += 2.0 * NFOPP_PI_D;
            dth -= NFOPP_PI_D;
            const double mean = prevp[2] + dth / 2.0;
            const bool fwd = cos(mean) * (p[0] - prevp[0]) + sin(mean) * (p[1] - prevp[1]) > 0.0;
            if (q == 1) fwd0 = fwd;
            else if (fwd != fwd0) {
              const int other = q - 1;
              if (other < 6) first = other > 1 ? other : 1;
              break;
            }
          }
          prevp[0] = p[0]; prevp[1] = p[1]; prevp[2] = p[2];
        }
      }
      if (count < 0) count = 0;
      s_first = first;
    }
    s_m = m;
    s_count = count;
  }
  __syncthreads();
  const int count = s_count;
  if (count < 0) {
    if (threadIdx.x == 0) a.count[b] = -1;
    return;
  }
  const int first = count > 0 ? s_first : 0;
  const int kept = count - first > 0 ? count - first : 0;
  if (threadIdx.x == 0) a.count[b] = kept;
  const int m = s_m;
  const double step = count > 1 ? 1.0 / (double)(count - 1) : 0.0;
  double* out = a.out + b * (long long)a.max_out * 3;
  for (int q = first + threadIdx.x; q < count && q - first < a.max_out; q += PP_THREADS) {
    const double x = (q == count - 1 && count > 1) ? 1.0 : (double)q * step;
    double p[3];
    spline_at(T, C, m, x, p);
    out[(q - first) * 3 + 0] = p[0];
    out[(q - first) * 3 + 1] = p[1];
    out[(q - first) * 3 + 2] = p[2];
  }
}

}  // namespace nfopp

using namespace nfopp;

extern "C" int nfopp_path_postprocess(const float* path_dev, int64_t batch, int32_t n_points, float minimal_distance,
                                      float distance_step, int32_t max_out, double* out_dev, int32_t* count_dev,
                                      void* stream) {
  NFOPP_REQUIRE(batch >= 0 && batch <= 0x7fffffffLL, "bad batch");
  NFOPP_REQUIRE(n_points >= 3 && n_points <= 1026, "path length must be 3..1026 poses (shorter paths pass through unchanged)");
  NFOPP_REQUIRE(distance_step > 0.f && minimal_distance >= 0.f && max_out >= 0, "bad distances");
  if (batch == 0) return NFOPP_OK;
  NFOPP_REQUIRE(path_dev && count_dev && (out_dev || max_out == 0), "null device pointer");
  PostArgs a;
  a.path = path_dev; a.n = n_points; a.max_out = max_out; a.min_dist = minimal_distance; a.dist_step = distance_step;
  a.out = out_dev; a.count = count_dev;
  // doubles: knots n+3, coefficients 3n, parameter n, pivots n, super-diagonal n; floats: poses 3n, lengths n
  const size_t lds = (size_t)(7 * n_points + 3) * 8 + (size_t)(4 * n_points) * 4;
  if (lds > 64 * 1024)
    NFOPP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(path_post_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(path_post_kernel, dim3((unsigned)batch), dim3(PP_THREADS), lds, (hipStream_t)stream, a);
  NFOPP_HIP(hipGetLastError());
  return NFOPP_OK;
}
