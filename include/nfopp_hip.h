/*
 * nfopp_hip.h -- C ABI of the MI355X-native NFOPP inner loop (libnfopp_hip.so).
 *
 * Drop-in boundary for ONE hot path of MisterMap/pytorch-motion-planner: the per-step work of
 * NERFOptPlanner / ConstrainedNERFOptPlanner `.step()`.  The reference has no native interface for this path
 * (it is PyTorch-CPU eager + autograd); each entry point below replaces the PyTorch op sequence cited next to
 * it (file:line under the reference root, `nfop/` = neural_field_optimal_planner/).  All pointers named *_dev are
 * DEVICE pointers (HIP, gfx950) to contiguous fp32 row-major arrays; sizes are element counts unless they say
 * bytes; `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are asynchronous on `stream`.
 * Every function returns 0 on success and a negative nfopp_status otherwise; nfopp_last_error() describes the
 * last failure of the calling thread.  No torch types, no ownership transfer: buffers are borrowed for the call.
 *
 * A batch of B trajectories is B independent reference problems that share one ONF (occupancy neural field).
 */
#ifndef NFOPP_HIP_H
#define NFOPP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NFOPP_ABI_VERSION 6
#define NFOPP_HIDDEN 100 /* width of both hidden layers, nfop/onf_model.py:18-23 */

typedef enum nfopp_status {
  NFOPP_OK = 0,
  NFOPP_ERR_ARG = -1,     /* bad shape / null pointer / unsupported configuration */
  NFOPP_ERR_HIP = -2,     /* HIP runtime error (message holds hipGetErrorString) */
  NFOPP_ERR_NO_DEVICE = -3
} nfopp_status;

/* Shape and normalisation of an ONF instance: nfop/onf_model.py:8-31.
 * The parameter buffer is ONE flat fp32 array in `state_dict()` order:
 *   [_angle_encoder._biases (2*angle_dim), _angle_encoder._frequencies (2*angle_dim)]   if angle_dim > 0
 *   mlp.0.weight [100, F], mlp.0.bias [100], mlp.2.weight [100,100], mlp.2.bias [100],
 *   mlp2.0.weight [1, 100+F], mlp2.0.bias [1], encoding_layer.weight [E, 2], [encoding_layer.bias [E]] if has_bias
 * with E = use_cos ? 200 : 100 and F = E + 2*angle_dim (220 for every shipped script). */
typedef struct nfopp_onf_config {
  float mean;         /* nfop/onf_model.py:38  x = (x - mean) / sigma */
  float sigma;
  int32_t use_cos;    /* 1: sin on the first 100 encodings, cos on the next 100 (onf_model.py:40-41) */
  int32_t has_bias;   /* encoding_layer bias present */
  int32_t angle_dim;  /* 0 = no AngleEncoder (2-D points), 10 = nfop/angle_encoder.py default */
} nfopp_onf_config;

/* Scalars of one trajectory optimisation step.  SE(2) terms: nfop/constrained_nerf_opt_planner.py:76-130,
 * boundary nfop/nerf_opt_planner.py:171-176, Adam = torch.optim.Adam single-tensor path, multiplier ascent
 * constrained:66-73.  adam_step_size = lr / (1 - beta1^k), adam_bc2_sqrt = sqrt(1 - beta2^k), both formed by the
 * caller in double precision for the 1-based step count k, exactly as torch does; adam_omb1/2 = 1 - beta1/2
 * likewise rounded from double (fp32(1 - 0.9) != 1.0f - 0.9f). */
typedef struct nfopp_traj_hyper {
  float collision_weight;
  float angle_weight;
  float constraint_deltas_weight;
  float multipliers_lr;
  float collision_multipliers_lr;
  float boundary_weight;
  float collision_beta;
  float direction_delta_weight;
  float bounds[4];          /* xmin, xmax, ymin, ymax */
  float adam_beta2, adam_omb1, adam_omb2, adam_eps;
  float adam_step_size, adam_bc2_sqrt;
} nfopp_traj_hyper;

#define NFOPP_NUM_TERMS 8 /* per-trajectory loss terms written by nfopp_traj_update (see below) */

int nfopp_abi_version(void);
const char* nfopp_last_error(void);
/* number of visible HIP devices (0 when none / no driver); never fails */
int nfopp_device_count(void);
/* number of fp32 parameters of an ONF with this configuration (33161 for the shipped one), <0 on bad config */
int64_t nfopp_onf_param_count(const nfopp_onf_config* cfg);

/* ONF forward + input gradient at explicit points.  Replaces `ONF.forward` (nfop/onf_model.py:33-50,
 * nfop/angle_encoder.py:15-18) followed by autograd w.r.t. the input.
 *   points_dev [P, point_dim]  point_dim = 3 (x, y, theta) when angle_dim > 0, else 2
 *   out4_dev   [P, 4]          logit, dlogit/dx, dlogit/dy, dlogit/dtheta (0 for 2-D fields) */
int nfopp_onf_eval_points(const nfopp_onf_config* cfg, const float* params_dev, const float* points_dev,
                          int64_t n_points, float* out4_dev, void* stream);

/* Forward only (the backward half of the kernel is skipped): out4_dev [P, 4] = logit, 0, 0, 0.  Used where the
 * reference evaluates the field without gradients (nfop/nerf_opt_planner.py:98-99,122-125, plotting). */
int nfopp_onf_eval_logits(const nfopp_onf_config* cfg, const float* params_dev, const float* points_dev,
                          int64_t n_points, float* out4_dev, void* stream);

/* Fused collision-point sampling + ONF forward + input gradient along a batch of trajectories: the ONF part
 * of `trajectory_loss` (constrained:78-85 for D = 3, nfop/nerf_opt_planner.py:113-117,157-169 for D = 2).
 *   traj_dev [B, N, D]   interior waypoints;  sample j of trajectory b lies between waypoints j and j+1
 *   t_dev    [B, N-1]    t_mode 0: read (injected draws);  t_mode 1: drawn here with Philox4x32-10
 *                        (key = seed, counter = (global sample index, rng_offset)) and WRITTEN for the
 *                        update kernel;  traj_index_offset = global index of trajectory 0 (multi-GPU shards
 *                        draw the same numbers as a single-GPU run)
 *   out4_dev [B, N-1, 4] as nfopp_onf_eval_points
 *   active_dev [B] uint8 or NULL (ABI 4): early stop, the reference's `break` (scripts/run_bench_mr.py:121-126).
 *                        Trajectories with active == 0 are compacted OUT of the sample stream, so the kernel's
 *                        work is proportional to the live ones; their t / out4 rows are left untouched and the
 *                        rows of live trajectories are bit-identical to a launch without the mask.
 *   live_ws_dev [B + 1] int32 workspace for the live list (required with active_dev, else may be NULL) */
int nfopp_traj_collision_eval(const nfopp_onf_config* cfg, const float* params_dev, const float* traj_dev,
                              int64_t batch, int32_t n_waypoints, int32_t dim, float* t_dev, int32_t t_mode,
                              uint64_t seed, uint64_t rng_offset, int64_t traj_index_offset, float* out4_dev,
                              const uint8_t* active_dev, int32_t* live_ws_dev, void* stream);

/* One `_optimize_trajectory` for every trajectory of the batch, given the ONF outputs at its samples:
 * loss terms + closed-form gradients (constrained:87-130, nerf:171-176), g <- H^-1 g (nerf:151), Adam
 * (nerf:154), multiplier ascent + clamp (constrained:66-73).  State arrays are updated IN PLACE.
 *   traj_dev [B,N,D], start_dev/goal_dev [B,D], lam_dev [B,N+1], cm_dev [B,N] (both NULL for D = 2),
 *   adam_m_dev/adam_v_dev [B,N,D], t_dev [B,N-1], onf_out4_dev [B,N-1,4]
 *   hinv_band_dev [2*half_width+1, N]: band of the reference's fp32 inverse Hessian (nerf:45-48), transposed so
 *       that entry [k][i] = Hinv[i][i + k - half_width] (0 outside the matrix)
 *   interior_lo/hi: waypoints i in [lo, hi) whose band column equals column lo BIT FOR BIT (the inverse of a
 *       tridiagonal Toeplitz matrix is Toeplitz away from the ends); their coefficients are broadcast from LDS.
 *       Pass lo = hi = 0 to disable.
 *   terms_dev [B, 8] or NULL: total, distance, sum softplus, sum lam*c, sum c^2, boundary, sum cm*tanh, sum relu(d)^2
 *   active_dev [B] uint8 or NULL: trajectories with 0 are left untouched (early stop, see nfopp_path_select_best) */
int nfopp_traj_update(const nfopp_traj_hyper* hp, int64_t batch, int32_t n_waypoints, int32_t dim,
                      float* traj_dev, const float* start_dev, const float* goal_dev, float* lam_dev,
                      float* cm_dev, float* adam_m_dev, float* adam_v_dev, const float* t_dev,
                      const float* onf_out4_dev, const float* hinv_band_dev, int32_t half_width,
                      int32_t interior_lo, int32_t interior_hi, float* terms_dev, const uint8_t* active_dev,
                      void* stream);

/* Arc-length reparametrisation (constrained:132-171 for D = 3 incl. multipliers; nerf:224-244 for D = 2).
 *   u_dev [N] = torch.linspace(0, 1, N+2)[1:-1] (formed by the caller so its rounding is the reference's) */
int nfopp_reparametrize(int64_t batch, int32_t n_waypoints, int32_t dim, float* traj_dev,
                        const float* start_dev, const float* goal_dev, float* lam_dev, float* cm_dev,
                        const float* u_dev, const uint8_t* active_dev, void* stream);

/* n planner steps of a FROZEN-field batch from one call (ABI 6): the step loops that drive the reference's hot path --
 * nfop/ros/goal_planner_adapter.py:50-52 (`while time < timeout: planner.step()`), scripts/run_planner.py:76-77,
 * scripts/run_bench_mr.py:109-132 -- without a host round trip per step.  Per step k = 0 .. n_steps-1, on `stream`:
 *   nfopp_traj_collision_eval (draws: t_mode 1 = Philox word rng_offset + k; t_mode 0 = row k of t_steps_dev [n_steps, B, N-1])
 *   nfopp_traj_update with the Adam scalars of step adam_steps_done + k + 1, formed here in double as torch.optim.Adam forms
 *     them (step_size = lr / (1 - beta1^k), bc2_sqrt = sqrt(1 - beta2^k); hp's two fields are ignored)
 *   nfopp_reparametrize when (step_count + k) % reparam_freq == 0           (nfop/nerf_opt_planner.py:60-71)
 * Same kernels and arguments as n single-step sequences: results are bit-identical to them.  terms_dev [B, 8] (or NULL)
 * receives the loss terms of the LAST step.  The caller advances its own counters by n_steps afterwards.  Nothing
 * synchronises.  Not for steps with ONF learning: the reference's fit needs the host checker between steps. */
typedef struct nfopp_traj_buffers {
  float* traj_dev;             /* [B, N, D] */
  const float* start_dev;      /* [B, D] */
  const float* goal_dev;       /* [B, D] */
  float* lam_dev;              /* [B, N+1], NULL for D = 2 */
  float* cm_dev;               /* [B, N],   NULL for D = 2 */
  float* adam_m_dev;           /* [B, N, D] */
  float* adam_v_dev;           /* [B, N, D] */
  float* t_dev;                /* [B, N-1] scratch for t_mode 1 (may be NULL for t_mode 0) */
  float* onf_out4_dev;         /* [B, N-1, 4] scratch */
  const float* hinv_band_dev;  /* [2*half_width+1, N] */
  const float* u_dev;          /* [N] = torch.linspace(0, 1, N+2)[1:-1] */
  const uint8_t* active_dev;   /* [B] or NULL */
  int32_t* live_ws_dev;        /* [B+1], required with active_dev */
  int64_t batch;
  int32_t n_waypoints, dim, half_width, interior_lo, interior_hi;
} nfopp_traj_buffers;

typedef struct nfopp_step_schedule {
  double adam_lr, adam_beta1, adam_beta2; /* the trajectory optimiser's Adam group (eps, 1-beta in nfopp_traj_hyper) */
  int64_t adam_steps_done;                /* Adam steps taken before this call */
  int64_t step_count;                     /* planner step counter before this call (reparametrisation schedule) */
  int64_t traj_index_offset;              /* global index of trajectory 0 (Philox counter) */
  uint64_t seed, rng_offset;              /* t_mode 1: Philox key, word of the first step */
  int32_t reparam_freq;                   /* >= 1 */
  int32_t t_mode;                         /* 0 = injected draws (t_steps_dev), 1 = in-kernel Philox */
} nfopp_step_schedule;

int nfopp_traj_steps(const nfopp_onf_config* cfg, const float* params_dev, const nfopp_traj_hyper* hp,
                     const nfopp_traj_buffers* buf, const nfopp_step_schedule* sched, int32_t n_steps,
                     const float* t_steps_dev, float* terms_dev, void* stream);

/* ONF fitting step, gradient part: BCE-with-logits (mean over ALL samples of the job) and its gradient w.r.t.
 * every parameter incl. the angle frequencies (nfop/nerf_opt_planner.py:83-89).
 *   samples_dev [P, point_dim], labels_dev [P] (0/1), inv_count = 1 / (global sample count)
 *   grad_dev [n_params + 2]: flat gradient in parameter order, then sum of per-sample losses * inv_count, then P
 *   workspace: nfopp_onf_train_workspace_bytes(cfg, P) bytes of device scratch
 * Reductions run in a fixed order (no float atomics): results are bitwise reproducible. */
size_t nfopp_onf_train_workspace_bytes(const nfopp_onf_config* cfg, int64_t n_samples);
int nfopp_onf_train_grad(const nfopp_onf_config* cfg, const float* params_dev, const float* samples_dev,
                         const float* labels_dev, int64_t n_samples, float inv_count, float* grad_dev,
                         void* workspace_dev, size_t workspace_bytes, void* stream);

/* Same, with an explicit implementation path: 0 = automatic (MFMA GEMM path from 2048 samples), 1 = per-sample
 * workgroups + thread-per-parameter reductions, 2 = MFMA forward/backward + split-K weight-gradient GEMMs. */
int nfopp_onf_train_grad_ex(const nfopp_onf_config* cfg, const float* params_dev, const float* samples_dev,
                            const float* labels_dev, int64_t n_samples, float inv_count, float* grad_dev,
                            void* workspace_dev, size_t workspace_bytes, int32_t path, void* stream);

/* ---- continuous ONF learning over a batch: ground truth and sample generation on the device --------------------
 * Ground-truth checkers (labels_dev[p] = 1.0 in collision, 0.0 free); bounds4 (host, may be NULL) = xmin,xmax,ymin,ymax
 * of nfop/collision_checker/collision_checker.py:12-19.
 *   circle:    any |pose.xy - obstacle| < radius                 nfop/collision_checker/circle_collision_checker.py:11-14
 *   rectangle: any obstacle inside box4 = (x0,x1,y0,y1) in the robot frame   .../rectangle_collision_checker.py:11-26
 *   grid:      uint8 occupancy image, cell = int((x - origin - cell/2)/cell) in float64 like the reference's numpy
 *              (geometry scalars are doubles since ABI 4); outside the image = collision
 *              (MapCollisionChecker, notebooks/onf_planner_image_map.ipynb cell 2; labels pinned by tests/golden/g16) */
int nfopp_check_collision_circle(const float* poses_dev, int64_t n, int32_t pose_dim, const float* obstacles_dev,
                                 int32_t n_obstacles, float radius, const float* bounds4, float* labels_dev,
                                 void* stream);
/* The circle checker with a uniform cell index over the obstacle points (same labels, far fewer distance tests):
 * obstacles_sorted_dev [n_obstacles, 2] sorted by cell (row-major cells of cell_size >= radius, origin cell_x0 / cell_y0),
 * cell_start_dev [cells_x * cells_y + 1] = first point of each cell. */
int nfopp_check_collision_circle_cells(const float* poses_dev, int64_t n, int32_t pose_dim,
                                       const float* obstacles_sorted_dev, int32_t n_obstacles,
                                       const int32_t* cell_start_dev, int32_t cells_x, int32_t cells_y, float cell_x0,
                                       float cell_y0, float cell_size, float radius, const float* bounds4,
                                       float* labels_dev, void* stream);
int nfopp_check_collision_rectangle(const float* poses_dev, int64_t n, const float* obstacles_dev, int32_t n_obstacles,
                                    const float* box4, const float* bounds4, float* labels_dev, void* stream);
int nfopp_check_collision_grid(const float* poses_dev, int64_t n, int32_t pose_dim, const uint8_t* grid_dev,
                               int32_t rows, int32_t cols, double origin_x, double origin_y, double cell_size,
                               float* labels_dev, void* stream);

/* Training-pose generation for every trajectory of a batch (nfop/nerf_opt_planner.py:101-120,135-141,
 * constrained:57-61,173-176).  Per trajectory: N-1 interpolated poses of the PREVIOUS trajectory -> "course" copies
 * (sigma course/angle) written to samples[b][0..N-2], "fine" copies appended to the candidate list behind the
 * retained pool (pool_count = 0 on the first step, pool_cap afterwards), n_field uniform field poses written to
 * samples[b][N-1+pool_cap ...].  Layouts: cand [B, pool_cap+N-1, D], samples [B, N-1+pool_cap+n_field, D].
 * Draws: Philox4x32-10 keyed by seed, counter (index, stream, trajectory, rng_offset). */
int nfopp_sample_candidates(const float* prev_traj_dev, int64_t batch, int32_t n_waypoints, int32_t dim,
                            int32_t pool_cap, int32_t pool_count, int32_t n_field, float course_sigma, float fine_sigma,
                            float angle_sigma, const float* bounds4, uint64_t seed, uint64_t rng_offset,
                            int64_t traj_index_offset, const float* pool_dev, const float* pool_age_dev,
                            float* cand_dev, float* cand_age_dev, float* samples_dev, void* stream);

/* Retained-pool resampling (nfop/nerf_opt_planner.py:122-133): weights sigmoid(logit)*exp(-0.03 age)+1e-6, pool_cap
 * candidates kept WITHOUT replacement with probability proportional to the weights (exponential race), ages + 1.
 * The first n_candidates of cand_stride (= pool_cap + N - 1) candidate slots per trajectory are valid;
 * onf_out4_dev [B, cand_stride, 4] = nfopp_onf_eval_points on cand_dev.  The new pool is also written to
 * samples[b][sample_offset ...] (sample_stride = poses per trajectory in the samples buffer). */
int nfopp_resample_pool(int64_t batch, int32_t n_candidates, int32_t cand_stride, int32_t pool_cap, int32_t dim,
                        int32_t sample_stride,
                        int32_t sample_offset, uint64_t seed, uint64_t rng_offset, int64_t traj_index_offset,
                        const float* cand_dev, const float* cand_age_dev, const float* onf_out4_dev, float* pool_dev,
                        float* pool_age_dev, float* samples_dev, void* stream);

/* ---- path evaluation (the step after the planner step: scripts/run_bench_mr.py:109-132) ---------------------------
 * nfopp_path_interpolate: densifies start -> waypoints -> goal with `sub` poses per segment (theta along the wrapped
 *   difference) into poses_dev [B, (N+1)*sub + 1, D] and writes the xy polyline length to length_dev [B].
 * The caller labels the poses with a ground-truth checker (above), then
 * nfopp_path_select_best: collides[b] = any label set; a collision-free path shorter than best_length[b] replaces
 *   best_traj[b]; a collision-free path that does not improve clears active[b] (the reference's `break`); inactive
 *   trajectories are skipped by nfopp_traj_update / nfopp_reparametrize when active_dev is passed to them. */
int nfopp_path_interpolate(const float* traj_dev, const float* start_dev, const float* goal_dev, int64_t batch,
                           int32_t n_waypoints, int32_t dim, int32_t sub, float* poses_dev, float* length_dev,
                           void* stream);
int nfopp_path_select_best(const float* labels_dev, const float* length_dev, const float* traj_dev, int64_t batch,
                           int32_t poses_per_path, int32_t n_waypoints, int32_t dim, float* best_traj_dev,
                           float* best_length_dev, uint8_t* collides_dev, uint8_t* active_dev, void* stream);
/* Matrix path of the fused ONF kernels (nfopp_onf_eval_points / _logits / nfopp_traj_collision_eval):
 *   1 (default) = bf16x3 split-precision MFMA: every fp32 operand is split EXACTLY into three bf16 levels and the six
 *       partial products above 2^-24 are accumulated in fp32 on the bf16 matrix pipe -- fp32-faithful (closer to float64
 *       than a sequential fp32 dot product).  Every launch of every ONF shape (F = 100 / 120 / 200 / 220) runs on 32x32x16 tiles
 *       (csrc/onf_x32_impl.h), in two workgroup shapes that are bit-identical per sample: results do not depend on the batch
 *       size or on how a batch is sharded; the training pass of the ONF fit too.
 *   0 = fp32 MFMA (v_mfma_f32_16x16x4_f32, csrc/onf_fused.hip).
 *   2 = bf16x3 split on the round-2 kernels (16x16x32 tiles, csrc/onf_split.hip) at every size: an independent implementation
 *       of path 1's arithmetic, kept as a cross-check.
 *   The environment variable NFOPP_MATRIX_PATH=fp32|0|1|2 selects the path at load time.  Process-wide.
 *   The split paths keep ONE scratch image per (device, stream) (pre-split weights, rebuilt from params_dev by a small
 *   kernel in front of every evaluation on the caller's stream), at most 16 per device: a 17th stream reuses the least
 *   recently used slot after a device synchronisation. */
int nfopp_set_matrix_path(int32_t path);
/* Content version of an ONF parameter buffer on the current device (ABI 5).  The split matrix paths evaluate the field from
 * a pre-split image of the weights that a small kernel rebuilds from params_dev in front of EVERY launch, because the
 * library cannot see whether the caller changed the buffer.  A caller who knows can vouch for it: register a non-zero
 * `version` that it changes whenever the buffer's contents change; while (buffer, version, configuration) stay the same,
 * launches on a stream reuse that stream's image and skip the rebuild (a frozen field: one launch less per planner step).
 * version = 0 withdraws the registration; nfopp_adam_step withdraws it for the buffer it updates.  Versions must be unique
 * per content over the life of the process (a counter), not per buffer. */
int nfopp_onf_params_version(const float* params_dev, uint64_t version);
int nfopp_get_matrix_path(void);

/* ---- the steps either side of the planner step ----------------------------------------
 * nfopp_init_trajectories: TrajectoryInitializer.initialize_trajectory (+ initialize_angle and, with
 *   angles_with_direction != 0, initialize_angle_with_trajectory_direction; nfop/trajectory_initializer.py:12-45) for
 *   a batch: traj_dev [B, N, D] <- straight line start -> goal with torch.linspace's fp32 rounding, theta along the
 *   wrapped shortest rotation, optionally pulled towards the travel direction by a 0 -> 1 -> 0 ramp.
 * nfopp_path_postprocess: PathPostprocessor.process (nfop/ros/path_postprocessor.py:13-69) for a batch of fp32 paths
 *   path_dev [B, n_points, 3] (3 <= n_points <= 1026): near-duplicate filter, quadratic-spline re-sampling every
 *   distance_step metres over the chord-length parameter (float64), leading direction flip trimmed.  count_dev[b] =
 *   poses path b produces (may exceed max_out: only the first max_out are written to out_dev [B, max_out, 3] float64;
 *   call with max_out = 0 to size the buffer), -1 when fewer than 3 poses survive the filter (the reference raises). */
int nfopp_init_trajectories(const float* start_dev, const float* goal_dev, int64_t batch, int32_t n_waypoints,
                            int32_t dim, int32_t angles_with_direction, float* traj_dev, void* stream);
int nfopp_path_postprocess(const float* path_dev, int64_t batch, int32_t n_points, float minimal_distance,
                           float distance_step, int32_t max_out, double* out_dev, int32_t* count_dev, void* stream);
/* torch.optim.Adam single-tensor update on a flat buffer (used for the ONF weights after the gradient
 * all-reduce): m.lerp_(g, 1-b1); v = b2 v + (1-b2) g^2; p -= step_size * m / (sqrt(v)/bc2_sqrt + eps). */
int nfopp_adam_step(float* param_dev, const float* grad_dev, float* m_dev, float* v_dev, int64_t n, float beta2,
                    float omb1, float omb2, float eps, float step_size, float bc2_sqrt, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NFOPP_HIP_H */
