// nfopp_traj_steps: n planner steps of a frozen-field batch enqueued from ONE call (ABI 6).
//
// The callers of the reference's hot path are step LOOPS -- nfop/ros/goal_planner_adapter.py:50-52
// (`while time < timeout: planner.step()`), scripts/run_planner.py:76-77, scripts/run_bench_mr.py:109-132 -- and at
// B = 1 one step is ~45 us of kernels: a Python round trip per step (argument marshalling, 2-3 ctypes calls, the
// hyper-parameter struct rebuilt with the step's Adam bias corrections) costs more than the kernels.  This entry point
// keeps the loop on the C side of the boundary: per step it forms the Adam scalars exactly as torch.optim.Adam does
// (Python / C doubles: 1 - beta^k through pow(), lr / bc1, sqrt(bc2), rounded to fp32 once) and enqueues
//   nfopp_traj_collision_eval -> nfopp_traj_update -> [nfopp_reparametrize when step_count % reparam_freq == 0]
// on the caller's stream -- the same three entry points, the same kernels, the same arguments as n single calls, so
// the results are bit-identical to n calls of the single-step sequence (tests/test_gpu_multistep.py).  Nothing
// synchronises; the field must not change during the n steps (the reference's step with ONF learning needs the host
// collision checker between steps and cannot be batched this way).
#include <math.h>

#include "common.h"

using namespace nfopp;

extern "C" int nfopp_traj_steps(const nfopp_onf_config* cfg, const float* params_dev, const nfopp_traj_hyper* hp,
                                const nfopp_traj_buffers* buf, const nfopp_step_schedule* sched, int32_t n_steps,
                                const float* t_steps_dev, float* terms_dev, void* stream) {
  NFOPP_REQUIRE(cfg && params_dev && hp && buf && sched, "null argument");
  NFOPP_REQUIRE(n_steps >= 0, "negative step count");
  NFOPP_REQUIRE(sched->reparam_freq >= 1, "reparam_freq must be >= 1 (pass a value beyond the step count to disable)");
  NFOPP_REQUIRE(sched->t_mode == 0 || sched->t_mode == 1, "t_mode must be 0 (injected draws) or 1 (in-kernel Philox)");
  NFOPP_REQUIRE(sched->t_mode == 1 || t_steps_dev || n_steps == 0, "t_mode 0 needs t_steps_dev [n_steps, B, N-1]");
  NFOPP_REQUIRE(sched->adam_steps_done >= 0 && sched->step_count >= 0, "negative step counters");
  NFOPP_REQUIRE(buf->dim == 3 || (buf->lam_dev == nullptr && buf->cm_dev == nullptr), "2-D trajectories carry no multipliers");
  NFOPP_REQUIRE(buf->u_dev, "u_dev (the reparametrisation grid) is required");
  nfopp_traj_hyper h = *hp;
  const int64_t row = buf->batch * (int64_t)(buf->n_waypoints - 1);
  for (int32_t k = 0; k < n_steps; ++k) {
    // torch.optim.Adam (single-tensor path): bias corrections in double for the 1-based step count
    const double step = (double)(sched->adam_steps_done + k + 1);
    const double bc1 = 1.0 - pow(sched->adam_beta1, step);
    const double bc2 = 1.0 - pow(sched->adam_beta2, step);
    h.adam_step_size = (float)(sched->adam_lr / bc1);
    h.adam_bc2_sqrt = (float)sqrt(bc2);
    float* t_k = sched->t_mode == 0 ? const_cast<float*>(t_steps_dev) + (int64_t)k * row : buf->t_dev;
    NFOPP_REQUIRE(t_k, "t_dev scratch [B, N-1] is required for t_mode 1");
    int rc = nfopp_traj_collision_eval(cfg, params_dev, buf->traj_dev, buf->batch, buf->n_waypoints, buf->dim, t_k,
                                       sched->t_mode, sched->seed, sched->rng_offset + (uint64_t)k, sched->traj_index_offset,
                                       buf->onf_out4_dev, buf->active_dev, buf->live_ws_dev, stream);
    if (rc != NFOPP_OK) return rc;
    rc = nfopp_traj_update(&h, buf->batch, buf->n_waypoints, buf->dim, buf->traj_dev, buf->start_dev, buf->goal_dev,
                           buf->lam_dev, buf->cm_dev, buf->adam_m_dev, buf->adam_v_dev, t_k, buf->onf_out4_dev,
                           buf->hinv_band_dev, buf->half_width, buf->interior_lo, buf->interior_hi,
                           k == n_steps - 1 ? terms_dev : nullptr, buf->active_dev, stream);
    if (rc != NFOPP_OK) return rc;
    if ((sched->step_count + k) % sched->reparam_freq == 0) {
      rc = nfopp_reparametrize(buf->batch, buf->n_waypoints, buf->dim, buf->traj_dev, buf->start_dev, buf->goal_dev,
                               buf->lam_dev, buf->cm_dev, buf->u_dev, buf->active_dev, stream);
      if (rc != NFOPP_OK) return rc;
    }
  }
  return NFOPP_OK;
}
