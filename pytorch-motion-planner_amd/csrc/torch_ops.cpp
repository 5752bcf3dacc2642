// torch.ops.nfopp.* -- the PyTorch-ROCm extension form of the drop-in boundary (last row; BASELINE
// north_star: "exposed as a PyTorch-ROCm extension").  A thin TORCH_LIBRARY shim over the C ABI of include/nfopp_hip.h:
// every op checks device / dtype / contiguity / shapes with TORCH_CHECK at the op boundary, takes the CURRENT HIP stream of
// the tensors' device, and calls the extern "C" entry point -- no arithmetic lives here.  Built as
// nfopp/lib/libnfopp_torch.so (links libnfopp_hip.so from the same directory); loaded with torch.ops.load_library.
//
// Reference side these ops stand in for: the autograd graph of ONF.forward (nfop/onf_model.py:33-50), one
// `_optimize_trajectory` (nfop/nerf_opt_planner.py:143-155, nfop/constrained_nerf_opt_planner.py:63-130), the
// reparametrisation (constrained:132-171, nerf:224-244) and one `_optimize_collision_model` step (nerf:76-91).
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>    // PyTorch-ROCm names its HIP devices "cuda": these are the
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>       // guard / stream types that accept that device type
#include <torch/library.h>

#include "nfopp_hip.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<Tensor>;

void check_tensor(const Tensor& t, const char* name, at::ScalarType dtype = at::kFloat) {
  TORCH_CHECK(t.is_cuda(), "nfopp: ", name, " must live on a HIP device (got ", t.device(), "); there is no CPU path");
  TORCH_CHECK(t.scalar_type() == dtype, "nfopp: ", name, " must be ", dtype, " (got ", t.scalar_type(), ")");
  TORCH_CHECK(t.is_contiguous(), "nfopp: ", name, " must be contiguous");
}
void same_device(const Tensor& a, const Tensor& b, const char* name) {
  TORCH_CHECK(a.device() == b.device(), "nfopp: ", name, " lives on ", b.device(), ", expected ", a.device());
}
template <class T>
T* opt_ptr(const OptTensor& t) {
  return (t.has_value() && t->defined() && t->numel() > 0) ? t->data_ptr<T>() : nullptr;
}
void check_status(int rc) { TORCH_CHECK(rc == 0, "nfopp call failed (", rc, "): ", nfopp_last_error()); }

nfopp_onf_config make_cfg(double mean, double sigma, bool use_cos, bool has_bias, int64_t angle_dim) {
  nfopp_onf_config c;
  c.mean = (float)mean; c.sigma = (float)sigma; c.use_cos = use_cos; c.has_bias = has_bias; c.angle_dim = (int32_t)angle_dim;
  return c;
}
void check_params(const Tensor& params, const nfopp_onf_config& c) {
  check_tensor(params, "params");
  const int64_t want = nfopp_onf_param_count(&c);
  TORCH_CHECK(want > 0, "nfopp: bad ONF configuration");
  TORCH_CHECK(params.numel() == want, "nfopp: params has ", params.numel(), " elements, this ONF configuration has ", want);
}
void* stream_of(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream(); }

// the 18 floats of nfopp_traj_hyper in declaration order
nfopp_traj_hyper make_hyper(at::ArrayRef<double> h) {
  TORCH_CHECK(h.size() == 18, "nfopp: hyper must hold the 18 floats of nfopp_traj_hyper (include/nfopp_hip.h), got ", h.size());
  nfopp_traj_hyper o;
  float* f = reinterpret_cast<float*>(&o);
  static_assert(sizeof(nfopp_traj_hyper) == 18 * sizeof(float), "nfopp_traj_hyper layout");
  for (int k = 0; k < 18; ++k) f[k] = (float)h[k];
  return o;
}

// ONF.forward + autograd w.r.t. the input (nfop/onf_model.py:33-50): out [P, 4] = logit, d/dx, d/dy, d/dtheta
Tensor onf_fwd_bwd_input(const Tensor& params, const Tensor& points, double mean, double sigma, bool use_cos, bool has_bias,
                         int64_t angle_dim) {
  const nfopp_onf_config c = make_cfg(mean, sigma, use_cos, has_bias, angle_dim);
  check_params(params, c);
  check_tensor(points, "points");
  same_device(params, points, "points");
  const int64_t dim = angle_dim > 0 ? 3 : 2;
  TORCH_CHECK(points.dim() == 2 && points.size(1) == dim, "nfopp: points must be [P, ", dim, "]");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(params.device());
  Tensor out = at::empty({points.size(0), 4}, points.options());
  check_status(nfopp_onf_eval_points(&c, params.data_ptr<float>(), points.data_ptr<float>(), points.size(0),
                                     out.data_ptr<float>(), stream_of(params)));
  return out;
}

// forward only (nfop/nerf_opt_planner.py:98-99,122-125): out [P, 1]
Tensor onf_logits(const Tensor& params, const Tensor& points, double mean, double sigma, bool use_cos, bool has_bias,
                  int64_t angle_dim) {
  const nfopp_onf_config c = make_cfg(mean, sigma, use_cos, has_bias, angle_dim);
  check_params(params, c);
  check_tensor(points, "points");
  same_device(params, points, "points");
  const int64_t dim = angle_dim > 0 ? 3 : 2;
  TORCH_CHECK(points.dim() == 2 && points.size(1) == dim, "nfopp: points must be [P, ", dim, "]");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(params.device());
  Tensor out = at::empty({points.size(0), 4}, points.options());
  check_status(nfopp_onf_eval_logits(&c, params.data_ptr<float>(), points.data_ptr<float>(), points.size(0),
                                     out.data_ptr<float>(), stream_of(params)));
  return out.narrow(1, 0, 1);
}

// one `_optimize_trajectory` for a batch: collision sampling + ONF (nfopp_traj_collision_eval), then losses, H^-1 g, Adam and
// the multiplier ascent (nfopp_traj_update).  State tensors are updated in place.
void traj_step(const Tensor& params, double mean, double sigma, bool use_cos, bool has_bias, int64_t angle_dim, Tensor traj,
               const Tensor& start, const Tensor& goal, const OptTensor& lam, const OptTensor& cm, Tensor adam_m, Tensor adam_v,
               Tensor t, int64_t t_mode, int64_t seed, int64_t rng_offset, int64_t traj_index_offset, Tensor onf_out,
               const Tensor& hinv_band, int64_t half_width, int64_t interior_lo, int64_t interior_hi, at::ArrayRef<double> hyper,
               const OptTensor& terms, const OptTensor& active, const OptTensor& live_ws) {
  const nfopp_onf_config c = make_cfg(mean, sigma, use_cos, has_bias, angle_dim);
  check_params(params, c);
  check_tensor(traj, "traj");
  TORCH_CHECK(traj.dim() == 3, "nfopp: traj must be [B, N, D]");
  const int64_t B = traj.size(0), N = traj.size(1), D = traj.size(2);
  TORCH_CHECK(D == (angle_dim > 0 ? 3 : 2), "nfopp: trajectory dim ", D, " does not match the ONF point dim");
  TORCH_CHECK(N >= 2, "nfopp: need at least 2 waypoints");
  auto need = [&](const Tensor& x, const char* name, at::IntArrayRef shape) {
    check_tensor(x, name);
    same_device(traj, x, name);
    TORCH_CHECK(x.sizes() == shape, "nfopp: ", name, " must have shape ", shape, ", got ", x.sizes());
  };
  same_device(traj, params, "params");
  need(start, "start", {B, D}); need(goal, "goal", {B, D});
  need(adam_m, "adam_m", {B, N, D}); need(adam_v, "adam_v", {B, N, D});
  need(t, "t", {B, N - 1}); need(onf_out, "onf_out", {B, N - 1, 4});
  check_tensor(hinv_band, "hinv_band");
  same_device(traj, hinv_band, "hinv_band");
  TORCH_CHECK(hinv_band.dim() == 2 && hinv_band.size(0) == 2 * half_width + 1 && hinv_band.size(1) == N,
              "nfopp: hinv_band must be [2 * half_width + 1, N]");
  if (D == 3) {
    TORCH_CHECK(lam.has_value() && cm.has_value(), "nfopp: the SE(2) step needs the multiplier tensors lam [B, N+1], cm [B, N]");
    need(*lam, "lam", {B, N + 1}); need(*cm, "cm", {B, N});
  } else {
    // the 2-D step has no multipliers: a tensor passed here would reach the kernel unvalidated
    TORCH_CHECK(!lam.has_value() && !cm.has_value(), "nfopp: the 2-D step takes no multiplier tensors (lam / cm must be None)");
  }
  if (terms.has_value()) need(*terms, "terms", {B, NFOPP_NUM_TERMS});
  if (active.has_value()) {
    check_tensor(*active, "active", at::kByte);
    same_device(traj, *active, "active");
    TORCH_CHECK(active->numel() == B, "nfopp: active must be [B] uint8");
    TORCH_CHECK(live_ws.has_value(), "nfopp: an active mask needs the live-list workspace (B + 1 int32)");
    check_tensor(*live_ws, "live_ws", at::kInt);
    same_device(traj, *live_ws, "live_ws");
    TORCH_CHECK(live_ws->numel() >= B + 1, "nfopp: live_ws must hold B + 1 int32");
  } else {
    TORCH_CHECK(!live_ws.has_value(), "nfopp: live_ws without an active mask");
  }
  const nfopp_traj_hyper hp = make_hyper(hyper);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(traj.device());
  void* st = stream_of(traj);
  check_status(nfopp_traj_collision_eval(&c, params.data_ptr<float>(), traj.data_ptr<float>(), B, (int32_t)N, (int32_t)D,
                                         t.data_ptr<float>(), (int32_t)t_mode, (uint64_t)seed, (uint64_t)rng_offset,
                                         traj_index_offset, onf_out.data_ptr<float>(), opt_ptr<uint8_t>(active),
                                         opt_ptr<int32_t>(live_ws), st));
  check_status(nfopp_traj_update(&hp, B, (int32_t)N, (int32_t)D, traj.data_ptr<float>(), start.data_ptr<float>(),
                                 goal.data_ptr<float>(), opt_ptr<float>(lam), opt_ptr<float>(cm), adam_m.data_ptr<float>(),
                                 adam_v.data_ptr<float>(), t.data_ptr<float>(), onf_out.data_ptr<float>(),
                                 hinv_band.data_ptr<float>(), (int32_t)half_width, (int32_t)interior_lo, (int32_t)interior_hi,
                                 opt_ptr<float>(terms), opt_ptr<uint8_t>(active), st));
}

// n frozen-field planner steps from one call (nfopp_traj_steps, ABI 6): the callers' step loops
// (nfop/ros/goal_planner_adapter.py:50-52, scripts/run_planner.py:76-77) without a host round trip per step.
// t_steps: [n_steps, B, N-1] injected draws (t_mode 0) or None (t_mode 1, in-kernel Philox; `t` is the scratch row).
void traj_steps(const Tensor& params, double mean, double sigma, bool use_cos, bool has_bias, int64_t angle_dim, Tensor traj,
                const Tensor& start, const Tensor& goal, const OptTensor& lam, const OptTensor& cm, Tensor adam_m, Tensor adam_v,
                Tensor t, const OptTensor& t_steps, int64_t seed, int64_t rng_offset, int64_t traj_index_offset, Tensor onf_out,
                const Tensor& hinv_band, int64_t half_width, int64_t interior_lo, int64_t interior_hi, const Tensor& u,
                at::ArrayRef<double> hyper, double adam_lr, double adam_beta1, double adam_beta2, int64_t adam_steps_done,
                int64_t step_count, int64_t reparam_freq, int64_t n_steps, const OptTensor& terms, const OptTensor& active,
                const OptTensor& live_ws) {
  const nfopp_onf_config c = make_cfg(mean, sigma, use_cos, has_bias, angle_dim);
  check_params(params, c);
  check_tensor(traj, "traj");
  TORCH_CHECK(traj.dim() == 3, "nfopp: traj must be [B, N, D]");
  const int64_t B = traj.size(0), N = traj.size(1), D = traj.size(2);
  TORCH_CHECK(D == (angle_dim > 0 ? 3 : 2), "nfopp: trajectory dim ", D, " does not match the ONF point dim");
  TORCH_CHECK(N >= 2, "nfopp: need at least 2 waypoints");
  TORCH_CHECK(n_steps >= 0 && reparam_freq >= 1 && adam_steps_done >= 0 && step_count >= 0, "nfopp: bad step schedule");
  auto need = [&](const Tensor& x, const char* name, at::IntArrayRef shape) {
    check_tensor(x, name);
    same_device(traj, x, name);
    TORCH_CHECK(x.sizes() == shape, "nfopp: ", name, " must have shape ", shape, ", got ", x.sizes());
  };
  same_device(traj, params, "params");
  need(start, "start", {B, D}); need(goal, "goal", {B, D});
  need(adam_m, "adam_m", {B, N, D}); need(adam_v, "adam_v", {B, N, D});
  need(t, "t", {B, N - 1}); need(onf_out, "onf_out", {B, N - 1, 4}); need(u, "u", {N});
  check_tensor(hinv_band, "hinv_band");
  same_device(traj, hinv_band, "hinv_band");
  TORCH_CHECK(hinv_band.dim() == 2 && hinv_band.size(0) == 2 * half_width + 1 && hinv_band.size(1) == N,
              "nfopp: hinv_band must be [2 * half_width + 1, N]");
  if (D == 3) {
    TORCH_CHECK(lam.has_value() && cm.has_value(), "nfopp: the SE(2) step needs the multiplier tensors lam [B, N+1], cm [B, N]");
    need(*lam, "lam", {B, N + 1}); need(*cm, "cm", {B, N});
  } else {
    TORCH_CHECK(!lam.has_value() && !cm.has_value(), "nfopp: the 2-D step takes no multiplier tensors (lam / cm must be None)");
  }
  if (t_steps.has_value()) need(*t_steps, "t_steps", {n_steps, B, N - 1});
  if (terms.has_value()) need(*terms, "terms", {B, NFOPP_NUM_TERMS});
  if (active.has_value()) {
    check_tensor(*active, "active", at::kByte);
    same_device(traj, *active, "active");
    TORCH_CHECK(active->numel() == B, "nfopp: active must be [B] uint8");
    TORCH_CHECK(live_ws.has_value(), "nfopp: an active mask needs the live-list workspace (B + 1 int32)");
    check_tensor(*live_ws, "live_ws", at::kInt);
    same_device(traj, *live_ws, "live_ws");
    TORCH_CHECK(live_ws->numel() >= B + 1, "nfopp: live_ws must hold B + 1 int32");
  } else {
    TORCH_CHECK(!live_ws.has_value(), "nfopp: live_ws without an active mask");
  }
  const nfopp_traj_hyper hp = make_hyper(hyper);
  nfopp_traj_buffers buf;
  buf.traj_dev = traj.data_ptr<float>(); buf.start_dev = start.data_ptr<float>(); buf.goal_dev = goal.data_ptr<float>();
  buf.lam_dev = opt_ptr<float>(lam); buf.cm_dev = opt_ptr<float>(cm);
  buf.adam_m_dev = adam_m.data_ptr<float>(); buf.adam_v_dev = adam_v.data_ptr<float>();
  buf.t_dev = t.data_ptr<float>(); buf.onf_out4_dev = onf_out.data_ptr<float>();
  buf.hinv_band_dev = hinv_band.data_ptr<float>(); buf.u_dev = u.data_ptr<float>();
  buf.active_dev = opt_ptr<uint8_t>(active); buf.live_ws_dev = opt_ptr<int32_t>(live_ws);
  buf.batch = B; buf.n_waypoints = (int32_t)N; buf.dim = (int32_t)D; buf.half_width = (int32_t)half_width;
  buf.interior_lo = (int32_t)interior_lo; buf.interior_hi = (int32_t)interior_hi;
  nfopp_step_schedule sc;
  sc.adam_lr = adam_lr; sc.adam_beta1 = adam_beta1; sc.adam_beta2 = adam_beta2;
  sc.adam_steps_done = adam_steps_done; sc.step_count = step_count; sc.traj_index_offset = traj_index_offset;
  sc.seed = (uint64_t)seed; sc.rng_offset = (uint64_t)rng_offset; sc.reparam_freq = (int32_t)reparam_freq;
  sc.t_mode = t_steps.has_value() ? 0 : 1;
  c10::hip::HIPGuardMasqueradingAsCUDA guard(traj.device());
  check_status(nfopp_traj_steps(&c, params.data_ptr<float>(), &hp, &buf, &sc, (int32_t)n_steps, opt_ptr<float>(t_steps),
                                opt_ptr<float>(terms), stream_of(traj)));
}

// arc-length reparametrisation (constrained:132-171 / nerf:224-244), in place
void reparametrize(Tensor traj, const Tensor& start, const Tensor& goal, const OptTensor& lam, const OptTensor& cm,
                   const Tensor& u, const OptTensor& active) {
  check_tensor(traj, "traj");
  TORCH_CHECK(traj.dim() == 3, "nfopp: traj must be [B, N, D]");
  const int64_t B = traj.size(0), N = traj.size(1), D = traj.size(2);
  check_tensor(start, "start"); check_tensor(goal, "goal"); check_tensor(u, "u");
  same_device(traj, start, "start"); same_device(traj, goal, "goal"); same_device(traj, u, "u");
  TORCH_CHECK(start.numel() == B * D && goal.numel() == B * D, "nfopp: start / goal must be [B, D]");
  TORCH_CHECK(u.numel() == N, "nfopp: u must be torch.linspace(0, 1, N + 2)[1:-1]");
  if (D == 3) {
    TORCH_CHECK(lam.has_value() && cm.has_value(), "nfopp: the SE(2) reparametrisation needs lam [B, N+1] and cm [B, N]");
    check_tensor(*lam, "lam"); check_tensor(*cm, "cm");
    same_device(traj, *lam, "lam"); same_device(traj, *cm, "cm");
    TORCH_CHECK(lam->numel() == B * (N + 1) && cm->numel() == B * N, "nfopp: lam must be [B, N+1], cm [B, N]");
  } else {
    TORCH_CHECK(D == 2, "nfopp: trajectory dim must be 2 or 3");
    TORCH_CHECK(!lam.has_value() && !cm.has_value(), "nfopp: the 2-D reparametrisation takes no multiplier tensors");
  }
  if (active.has_value()) {
    check_tensor(*active, "active", at::kByte);
    same_device(traj, *active, "active");
    TORCH_CHECK(active->numel() == B, "nfopp: active must be [B] uint8");
  }
  c10::hip::HIPGuardMasqueradingAsCUDA guard(traj.device());
  check_status(nfopp_reparametrize(B, (int32_t)N, (int32_t)D, traj.data_ptr<float>(), start.data_ptr<float>(),
                                   goal.data_ptr<float>(), opt_ptr<float>(lam), opt_ptr<float>(cm), u.data_ptr<float>(),
                                   opt_ptr<uint8_t>(active), stream_of(traj)));
}

// gradient of the BCE-with-logits fitting loss w.r.t. every ONF parameter (nerf:83-89): [n_params | loss | count]
Tensor onf_train_grad(const Tensor& params, const Tensor& samples, const Tensor& labels, double inv_count, double mean,
                      double sigma, bool use_cos, bool has_bias, int64_t angle_dim) {
  const nfopp_onf_config c = make_cfg(mean, sigma, use_cos, has_bias, angle_dim);
  check_params(params, c);
  check_tensor(samples, "samples"); check_tensor(labels, "labels");
  same_device(params, samples, "samples"); same_device(params, labels, "labels");
  const int64_t dim = angle_dim > 0 ? 3 : 2;
  TORCH_CHECK(samples.dim() == 2 && samples.size(1) == dim, "nfopp: samples must be [P, ", dim, "]");
  TORCH_CHECK(labels.numel() == samples.size(0), "nfopp: labels must be [P]");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(params.device());
  const int64_t P = samples.size(0);
  const size_t ws_bytes = nfopp_onf_train_workspace_bytes(&c, P);
  Tensor ws = at::empty({(int64_t)((ws_bytes + 3) / 4)}, params.options());
  Tensor grad = at::empty({params.numel() + 2}, params.options());
  check_status(nfopp_onf_train_grad(&c, params.data_ptr<float>(), samples.data_ptr<float>(), labels.data_ptr<float>(), P,
                                    (float)inv_count, grad.data_ptr<float>(), ws.data_ptr<float>(), ws_bytes, stream_of(params)));
  return grad;
}

// torch.optim.Adam single-tensor update on the flat parameter buffer (nerf:90), in place
void adam_step(Tensor param, const Tensor& grad, Tensor m, Tensor v, double beta2, double omb1, double omb2, double eps,
               double step_size, double bc2_sqrt) {
  check_tensor(param, "param"); check_tensor(grad, "grad"); check_tensor(m, "m"); check_tensor(v, "v");
  same_device(param, grad, "grad"); same_device(param, m, "m"); same_device(param, v, "v");
  const int64_t n = param.numel();
  TORCH_CHECK(grad.numel() >= n && m.numel() == n && v.numel() == n, "nfopp: grad / m / v must cover the ", n, " parameters");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(param.device());
  check_status(nfopp_adam_step(param.data_ptr<float>(), grad.data_ptr<float>(), m.data_ptr<float>(), v.data_ptr<float>(), n,
                               (float)beta2, (float)omb1, (float)omb2, (float)eps, (float)step_size, (float)bc2_sqrt,
                               stream_of(param)));
}

// one `_optimize_collision_model` step on one GPU: gradient + Adam (multi-GPU callers all-reduce the gradient in between:
// nfopp/batch.py OnfFitter).  Returns the gradient buffer (its [-2] entry is the loss).
Tensor onf_train_step(Tensor params, Tensor m, Tensor v, const Tensor& samples, const Tensor& labels, double mean, double sigma,
                      bool use_cos, bool has_bias, int64_t angle_dim, double beta2, double omb1, double omb2, double eps,
                      double step_size, double bc2_sqrt) {
  TORCH_CHECK(samples.dim() == 2 && samples.size(0) > 0, "nfopp: samples must be [P, D] with P > 0");
  Tensor grad = onf_train_grad(params, samples, labels, 1.0 / (double)samples.size(0), mean, sigma, use_cos, has_bias, angle_dim);
  adam_step(params, grad, m, v, beta2, omb1, omb2, eps, step_size, bc2_sqrt);
  return grad;
}

}  // namespace

TORCH_LIBRARY(nfopp, lib) {
  lib.def("onf_fwd_bwd_input(Tensor params, Tensor points, float mean, float sigma, bool use_cos, bool has_bias, int angle_dim) -> Tensor");
  lib.def("onf_logits(Tensor params, Tensor points, float mean, float sigma, bool use_cos, bool has_bias, int angle_dim) -> Tensor");
  lib.def(
      "traj_step(Tensor params, float mean, float sigma, bool use_cos, bool has_bias, int angle_dim, Tensor(a!) traj, Tensor start, "
      "Tensor goal, Tensor(b!)? lam, Tensor(c!)? cm, Tensor(d!) adam_m, Tensor(e!) adam_v, Tensor(f!) t, int t_mode, int seed, "
      "int rng_offset, int traj_index_offset, Tensor(g!) onf_out, Tensor hinv_band, int half_width, int interior_lo, int interior_hi, "
      "float[] hyper, Tensor(h!)? terms, Tensor? active, Tensor(i!)? live_ws) -> ()");
  lib.def(
      "traj_steps(Tensor params, float mean, float sigma, bool use_cos, bool has_bias, int angle_dim, Tensor(a!) traj, Tensor start, "
      "Tensor goal, Tensor(b!)? lam, Tensor(c!)? cm, Tensor(d!) adam_m, Tensor(e!) adam_v, Tensor(f!) t, Tensor? t_steps, int seed, "
      "int rng_offset, int traj_index_offset, Tensor(g!) onf_out, Tensor hinv_band, int half_width, int interior_lo, int interior_hi, "
      "Tensor u, float[] hyper, float adam_lr, float adam_beta1, float adam_beta2, int adam_steps_done, int step_count, "
      "int reparam_freq, int n_steps, Tensor(h!)? terms, Tensor? active, Tensor(i!)? live_ws) -> ()");
  lib.def("reparametrize(Tensor(a!) traj, Tensor start, Tensor goal, Tensor(b!)? lam, Tensor(c!)? cm, Tensor u, Tensor? active) -> ()");
  lib.def("onf_train_grad(Tensor params, Tensor samples, Tensor labels, float inv_count, float mean, float sigma, bool use_cos, "
          "bool has_bias, int angle_dim) -> Tensor");
  lib.def("adam_step(Tensor(a!) param, Tensor grad, Tensor(b!) m, Tensor(c!) v, float beta2, float omb1, float omb2, float eps, "
          "float step_size, float bc2_sqrt) -> ()");
  lib.def("onf_train_step(Tensor(a!) params, Tensor(b!) m, Tensor(c!) v, Tensor samples, Tensor labels, float mean, float sigma, "
          "bool use_cos, bool has_bias, int angle_dim, float beta2, float omb1, float omb2, float eps, float step_size, "
          "float bc2_sqrt) -> Tensor");
}

// The ops validate their arguments themselves (device included: a CPU tensor gets the "no CPU path" message instead of a
// dispatcher "no kernel" error), so they are registered for every dispatch key.
TORCH_LIBRARY_IMPL(nfopp, CompositeExplicitAutograd, lib) {
  lib.impl("onf_fwd_bwd_input", &onf_fwd_bwd_input);
  lib.impl("onf_logits", &onf_logits);
  lib.impl("traj_step", &traj_step);
  lib.impl("traj_steps", &traj_steps);
  lib.impl("reparametrize", &reparametrize);
  lib.impl("onf_train_grad", &onf_train_grad);
  lib.impl("adam_step", &adam_step);
  lib.impl("onf_train_step", &onf_train_step);
}
