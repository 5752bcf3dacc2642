// ONF fitting step, gradient part (continuous ONF learning).
//
// Replaces (reference, PyTorch-CPU + autograd): `_optimize_collision_model` nfop/nerf_opt_planner.py:83-89 --
// ONF.forward on the sampled poses, BCEWithLogitsLoss(mean), loss.backward() over ALL parameters (the reference
// calls `collision_model.requires_grad_(True)`, which also un-freezes the angle frequencies).  Closed-form
// gradients: SURVEY.md Appendix A "weight-gradient pass".
//
// Three stages, all reductions in a fixed order (bitwise reproducible, no float atomics):
//   1. onf_train_sample_kernel : one workgroup per sample; forward, upstream rho = (sigmoid(l) - y) * inv_count,
//      backward to every layer; the per-sample factors of each weight gradient go to the workspace
//   2. onf_train_reduce_kernel : grad[o] = sum_p A[p][a(o)] * B[p][b(o)] over one chunk of samples per blockIdx.y
//   3. onf_train_final_kernel  : sums the chunk partials (and the per-sample losses) in chunk order
// Version 1 is sized for the single-trajectory planner (P = N + 109 samples per step); the large-P path of
// BASELINE config 5 reuses it with more chunks (an MFMA split-K version is future work).
#include "onf_kernel.h"

namespace nfopp {

constexpr int TS_THREADS = 128;
constexpr int H = NFOPP_HIDDEN;
constexpr int MAX_CHUNKS = 64;

struct TrainWs {  // workspace carve-up (float offsets)
  long long in, h1, h2, dh2, dh1, de, dzb, dzf, u, rho, loss, partial, total;
  int n_chunks;
};

static TrainWs carve(const OnfGeom& g, long long P) {
  TrainWs w;
  long long o = 0;
  w.in = o; o += P * g.fin;
  w.h1 = o; o += P * H;
  w.h2 = o; o += P * H;
  w.dh2 = o; o += P * H;
  w.dh1 = o; o += P * H;
  w.de = o; o += P * g.n_enc;
  w.dzb = o; o += P * g.n_ang;
  w.dzf = o; o += P * g.n_ang;
  w.u = o; o += P * 2;
  w.rho = o; o += P;
  w.loss = o; o += P;
  long long chunks = (P + 63) / 64;
  w.n_chunks = (int)(chunks < 1 ? 1 : (chunks > MAX_CHUNKS ? MAX_CHUNKS : chunks));
  w.partial = o; o += (long long)w.n_chunks * g.n_params;
  w.total = o;
  return w;
}

struct TrainArgs {
  OnfGeom geom;
  const float* params;
  const float* samples;
  const float* labels;
  long long P;
  float inv_count;
  float* ws;
  TrainWs w;
  float* grad;
};

__global__ __launch_bounds__(TS_THREADS) void onf_train_sample_kernel(const TrainArgs a) {
  __shared__ float in_s[256], arg_s[256], h1_s[H], h2_s[H], a1_s[H], a2_s[H], dh_s[H], red[TS_THREADS];
  __shared__ float rho_s;
  const OnfGeom& g = a.geom;
  const float* P = a.params;
  const long long p = blockIdx.x;
  const int tid = threadIdx.x;
  const float* x = a.samples + p * g.point_dim;
  const float ux = (x[0] - g.mean) / g.sigma, uy = (x[1] - g.mean) / g.sigma;
  const float th = g.point_dim == 3 ? x[2] : 0.0f;
  float* ws = a.ws;

  for (int f = tid; f < g.fin; f += TS_THREADS) {
    float arg;
    int cosf_;
    if (f < g.n_enc) {
      const float b = g.off_be >= 0 ? P[g.off_be + f] : 0.0f;
      arg = fmaf(P[g.off_we + 2 * f], ux, fmaf(P[g.off_we + 2 * f + 1], uy, b));
      cosf_ = (g.n_enc > g.n_sin && f >= g.n_sin) ? 1 : 0;
    } else {
      const int k = f - g.n_enc;
      arg = (th + P[g.off_ang_b + k]) * P[g.off_ang_f + k];
      cosf_ = k >= g.ang_dim ? 1 : 0;
    }
    const float v = sin_quadrant(arg, cosf_);
    in_s[f] = v;
    arg_s[f] = arg;
    ws[a.w.in + p * g.fin + f] = v;
  }
  if (tid == 0) { ws[a.w.u + 2 * p] = ux; ws[a.w.u + 2 * p + 1] = uy; }
  __syncthreads();
  if (tid < H) {
    float acc = P[g.off_b1 + tid];
    const float* w = P + g.off_w1 + tid * g.fin;
    for (int k = 0; k < g.fin; ++k) acc = fmaf(w[k], in_s[k], acc);
    a1_s[tid] = acc;
    h1_s[tid] = fmaxf(acc, 0.0f);
    ws[a.w.h1 + p * H + tid] = h1_s[tid];
  }
  __syncthreads();
  if (tid < H) {
    float acc = P[g.off_b2 + tid];
    const float* w = P + g.off_w2 + tid * H;
    for (int k = 0; k < H; ++k) acc = fmaf(w[k], h1_s[k], acc);
    a2_s[tid] = acc;
    h2_s[tid] = fmaxf(acc, 0.0f);
    ws[a.w.h2 + p * H + tid] = h2_s[tid];
  }
  __syncthreads();
  // logit = w3 . [h2, in] + b3 : fixed-order tree over 128 partial sums
  {
    float part = 0.f;
    for (int j = tid; j < H + g.fin; j += TS_THREADS) part = fmaf(P[g.off_w3 + j], j < H ? h2_s[j] : in_s[j - H], part);
    red[tid] = part;
    __syncthreads();
    for (int s = TS_THREADS / 2; s > 0; s >>= 1) {
      if (tid < s) red[tid] += red[tid + s];
      __syncthreads();
    }
    if (tid == 0) {
      const float logit = red[0] + P[g.off_b3];
      const float y = a.labels[p];
      const float loss = fmaxf(logit, 0.0f) - logit * y + log1pf(expf(-fabsf(logit)));
      const float sig = 1.0f / (1.0f + expf(-logit));
      const float rho = (sig - y) * a.inv_count;
      rho_s = rho;
      ws[a.w.rho + p] = rho;
      ws[a.w.loss + p] = loss * a.inv_count;
    }
    __syncthreads();
  }
  const float rho = rho_s;
  if (tid < H) {
    const float d = a2_s[tid] > 0.0f ? rho * P[g.off_w3 + tid] : 0.0f;
    dh_s[tid] = d;
    ws[a.w.dh2 + p * H + tid] = d;
  }
  __syncthreads();
  float dh1 = 0.f;
  if (tid < H) {
    float acc = 0.f;
    for (int m = 0; m < H; ++m) acc = fmaf(P[g.off_w2 + m * H + tid], dh_s[m], acc);
    dh1 = a1_s[tid] > 0.0f ? acc : 0.0f;
    ws[a.w.dh1 + p * H + tid] = dh1;
  }
  __syncthreads();
  if (tid < H) dh_s[tid] = dh1;
  __syncthreads();
  for (int f = tid; f < g.fin; f += TS_THREADS) {
    float acc = rho * P[g.off_w3 + H + f];
    for (int m = 0; m < H; ++m) acc = fmaf(P[g.off_w1 + m * g.fin + f], dh_s[m], acc);
    if (f < g.n_enc) {
      const int cosf_ = (g.n_enc > g.n_sin && f >= g.n_sin) ? 1 : 0;
      ws[a.w.de + p * g.n_enc + f] = acc * sin_quadrant(arg_s[f], cosf_ + 1);
    } else {
      const int k = f - g.n_enc;
      const float dz = acc * sin_quadrant(arg_s[f], (k >= g.ang_dim ? 1 : 0) + 1);
      ws[a.w.dzb + p * g.n_ang + k] = dz * P[g.off_ang_f + k];
      ws[a.w.dzf + p * g.n_ang + k] = dz * (th + P[g.off_ang_b + k]);
    }
  }
}

// one gradient segment: out[o] = sum_p A[p*sa + o / nb] * (B ? B[p*sb + o % nb] : 1)
struct Segment {
  int out_off, count, nb;
  long long a_off, b_off;  // workspace offsets; b_off < 0 => no B factor
  int sa, sb;
};
constexpr int MAX_SEG = 12;
struct ReduceArgs {
  Segment seg[MAX_SEG];
  int n_seg, n_params, n_chunks;
  long long P;
  float* ws;
  long long partial_off, loss_off;
  float* grad;
};

__global__ __launch_bounds__(256) void onf_train_reduce_kernel(const ReduceArgs a) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= a.n_params) return;
  int s = 0;
  while (s + 1 < a.n_seg && o >= a.seg[s + 1].out_off) ++s;
  const Segment sg = a.seg[s];
  const int ol = o - sg.out_off;
  const int ia = ol / sg.nb, ib = ol - ia * sg.nb;
  const int chunk = blockIdx.y;
  const long long per = (a.P + a.n_chunks - 1) / a.n_chunks;
  const long long p0 = chunk * per, p1 = min(a.P, p0 + per);
  const float* A = a.ws + sg.a_off + ia;
  float acc = 0.f;
  if (sg.b_off >= 0) {
    const float* B = a.ws + sg.b_off + ib;
    for (long long p = p0; p < p1; ++p) acc = fmaf(A[p * sg.sa], B[p * sg.sb], acc);
  } else {
    for (long long p = p0; p < p1; ++p) acc += A[p * sg.sa];
  }
  a.ws[a.partial_off + (long long)chunk * a.n_params + o] = acc;
}

__global__ __launch_bounds__(256) void onf_train_final_kernel(const ReduceArgs a) {
  __shared__ float red[256];
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o < a.n_params) {
    float acc = 0.f;
    for (int c = 0; c < a.n_chunks; ++c) acc += a.ws[a.partial_off + (long long)c * a.n_params + o];
    a.grad[o] = acc;
  }
  if (blockIdx.x == 0) {  // mean BCE loss: per-sample losses summed in a fixed strided-tree order
    float part = 0.f;
    for (long long p = threadIdx.x; p < a.P; p += 256) part += a.ws[a.loss_off + p];
    red[threadIdx.x] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      a.grad[a.n_params] = red[0];
      a.grad[a.n_params + 1] = (float)a.P;
    }
  }
}

}  // namespace nfopp

using namespace nfopp;

static const long long MFMA_PATH_MIN_SAMPLES = 2048;  // below this the per-sample path is launch-bound anyway
static const long long PER_SAMPLE_MAX = 65536;        // largest fit the per-sample path is budgeted for (path = 1 on request)

static int train_grad_per_sample(const OnfGeom& g, const float* params_dev, const float* samples_dev,
                                 const float* labels_dev, int64_t n_samples, float inv_count, float* grad_dev,
                                 void* workspace_dev, hipStream_t st);

extern "C" size_t nfopp_onf_train_workspace_bytes(const nfopp_onf_config* cfg, int64_t n_samples) {
  OnfGeom g;
  if (!make_geom(cfg, &g) || n_samples < 0) return 0;
  // the per-sample path keeps 3.5 KB per sample: it serves small fits (and, on request, up to PER_SAMPLE_MAX samples);
  // beyond that only the MFMA path's compact factors (1.84 KB per sample) are budgeted
  const size_t b = wgrad_workspace_bytes(g, n_samples);
  if (n_samples > PER_SAMPLE_MAX) return b;
  const size_t a = (size_t)carve(g, n_samples).total * sizeof(float);
  return a > b ? a : b;
}

extern "C" int nfopp_onf_train_grad_ex(const nfopp_onf_config* cfg, const float* params_dev, const float* samples_dev,
                                       const float* labels_dev, int64_t n_samples, float inv_count, float* grad_dev,
                                       void* workspace_dev, size_t workspace_bytes, int32_t path, void* stream) {
  OnfGeom g;
  NFOPP_REQUIRE(make_geom(cfg, &g), "bad ONF configuration");
  NFOPP_REQUIRE(g.fin <= 256, "feature dimension %d too large", g.fin);
  NFOPP_REQUIRE(params_dev && samples_dev && labels_dev && grad_dev && workspace_dev, "null device pointer");
  NFOPP_REQUIRE(n_samples > 0 && n_samples <= 0x7fffffffLL, "sample count out of range");
  NFOPP_REQUIRE(path >= 0 && path <= 2, "path must be 0 (auto), 1 (per-sample) or 2 (MFMA)");
  NFOPP_REQUIRE(path != 1 || n_samples <= PER_SAMPLE_MAX, "the per-sample path takes at most %lld samples",
                (long long)PER_SAMPLE_MAX);
  NFOPP_REQUIRE(workspace_bytes >= nfopp_onf_train_workspace_bytes(cfg, n_samples),
                "workspace too small: %zu < %zu bytes", workspace_bytes, nfopp_onf_train_workspace_bytes(cfg, n_samples));
  const bool mfma = path == 2 || (path == 0 && n_samples >= MFMA_PATH_MIN_SAMPLES);
  if (mfma)
    return onf_train_grad_mfma(g, params_dev, samples_dev, labels_dev, n_samples, inv_count, grad_dev,
                               (float*)workspace_dev, (hipStream_t)stream);
  return train_grad_per_sample(g, params_dev, samples_dev, labels_dev, n_samples, inv_count, grad_dev, workspace_dev,
                               (hipStream_t)stream);
}

extern "C" int nfopp_onf_train_grad(const nfopp_onf_config* cfg, const float* params_dev, const float* samples_dev,
                                    const float* labels_dev, int64_t n_samples, float inv_count, float* grad_dev,
                                    void* workspace_dev, size_t workspace_bytes, void* stream) {
  return nfopp_onf_train_grad_ex(cfg, params_dev, samples_dev, labels_dev, n_samples, inv_count, grad_dev, workspace_dev,
                                 workspace_bytes, 0, stream);
}

static int train_grad_per_sample(const OnfGeom& g, const float* params_dev, const float* samples_dev,
                                 const float* labels_dev, int64_t n_samples, float inv_count, float* grad_dev,
                                 void* workspace_dev, hipStream_t st) {
  TrainArgs a = {};
  a.geom = g;
  a.w = carve(g, n_samples);
  a.params = params_dev; a.samples = samples_dev; a.labels = labels_dev; a.P = n_samples;
  a.inv_count = inv_count; a.ws = (float*)workspace_dev; a.grad = grad_dev;
  hipLaunchKernelGGL(onf_train_sample_kernel, dim3((unsigned)n_samples), dim3(TS_THREADS), 0, st, a);
  NFOPP_HIP(hipGetLastError());

  ReduceArgs r = {};
  int n = 0;
  auto add = [&](int off, int count, int nb, long long a_off, int sa, long long b_off, int sb) {
    if (off < 0 || count <= 0) return;
    r.seg[n].out_off = off; r.seg[n].count = count; r.seg[n].nb = nb;
    r.seg[n].a_off = a_off; r.seg[n].sa = sa; r.seg[n].b_off = b_off; r.seg[n].sb = sb;
    ++n;
  };
  // segments in parameter order (offsets ascending)
  add(g.off_ang_b, g.n_ang, 1, a.w.dzb, g.n_ang, -1, 0);
  add(g.off_ang_f, g.n_ang, 1, a.w.dzf, g.n_ang, -1, 0);
  add(g.off_w1, H * g.fin, g.fin, a.w.dh1, H, a.w.in, g.fin);
  add(g.off_b1, H, 1, a.w.dh1, H, -1, 0);
  add(g.off_w2, H * H, H, a.w.dh2, H, a.w.h1, H);
  add(g.off_b2, H, 1, a.w.dh2, H, -1, 0);
  // w3 = rho^T [h2, in]: two segments with A = rho (one column), B = h2 / in
  add(g.off_w3, H, H, a.w.rho, 1, a.w.h2, H);
  add(g.off_w3 + H, g.fin, g.fin, a.w.rho, 1, a.w.in, g.fin);
  add(g.off_b3, 1, 1, a.w.rho, 1, -1, 0);
  add(g.off_we, 2 * g.n_enc, 2, a.w.de, g.n_enc, a.w.u, 2);
  add(g.off_be, g.n_enc, 1, a.w.de, g.n_enc, -1, 0);
  NFOPP_REQUIRE(n <= MAX_SEG, "internal: too many gradient segments");
  r.n_seg = n; r.n_params = g.n_params; r.n_chunks = a.w.n_chunks; r.P = n_samples; r.ws = a.ws;
  r.partial_off = a.w.partial; r.loss_off = a.w.loss; r.grad = grad_dev;
  const unsigned gx = (unsigned)((g.n_params + 255) / 256);
  hipLaunchKernelGGL(onf_train_reduce_kernel, dim3(gx, (unsigned)r.n_chunks), dim3(256), 0, st, r);
  NFOPP_HIP(hipGetLastError());
  hipLaunchKernelGGL(onf_train_final_kernel, dim3(gx), dim3(256), 0, st, r);
  NFOPP_HIP(hipGetLastError());
  return NFOPP_OK;
}
