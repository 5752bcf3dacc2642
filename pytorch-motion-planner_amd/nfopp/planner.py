"""Drop-in planners for the reference's hot path, backed by the HIP pipeline.

`NERFOptPlanner` / `ConstrainedNERFOptPlanner` keep the reference's constructor signatures, public methods
(`init, step, get_path, set_boundaries, update_goal_point, update_start_point` -- nfop/continuous_planner.py:4-27)
and the de-facto attributes drivers poke (`_device, _collision_model, checked_positions, truth_collision,
_collision_positions, full_trajectory(), _start_point, _goal_point, _step_count`), for ONE trajectory, exactly like
the reference (nfop/nerf_opt_planner.py:10-248, nfop/constrained_nerf_opt_planner.py:12-194).

What runs where:
  * trajectory optimisation, reparametrisation, ONF evaluation and the ONF fitting step: HIP kernels (engine.py,
    C ABI include/nfopp_hip.h).  No autograd, no PyTorch arithmetic.
  * sampling of ONF training poses and the ground-truth collision checker: host numpy, because the checker is a
    user-supplied host object (`check_collision`) and the reference draws these poses from numpy's global RNG; the
    draws are made in the reference's call order, so equal seeds give equal sample sets.
`rng="reference"` draws the per-step interpolation parameters t from torch's CPU generator (SE(2)) / numpy (2-D)
like the reference; `rng="device"` uses the in-kernel Philox stream instead.
"""
import numpy as np
import torch

from . import _lib
from .engine import TrajectoryEngine, TrajectoryHyper
from .host_utils import Position2, TrajectoryInitializer
from .path_tools import init_trajectories


class ContinuousPlanner(object):
    """The six-method planner interface of the reference (nfop/continuous_planner.py:4-27)."""

    def init(self, start_point, goal_point, boundaries):
        raise NotImplementedError()

    def step(self):
        raise NotImplementedError()

    def get_path(self):
        raise NotImplementedError()

    def set_boundaries(self, boundaries):
        raise NotImplementedError()

    def update_goal_point(self, goal_point):
        raise NotImplementedError()

    def update_start_point(self, start_point):
        raise NotImplementedError()


def _adam_group(optimizer):
    g = optimizer.param_groups[0]
    return float(g["lr"]), tuple(float(b) for b in g["betas"]), float(g["eps"])


class NERFOptPlanner(ContinuousPlanner):
    """2-D planner (nfop/nerf_opt_planner.py:10-248): loss = sum |dq|^2 + w_col * sum softplus(ONF(p))."""

    point_dim = 2

    def __init__(self, trajectory, collision_model, collision_checker, collision_optimizer, trajectory_optimizer,
                 trajectory_random_offset, collision_weight, velocity_hessian_weight, init_collision_iteration=100,
                 init_collision_points=100, reparametrize_trajectory_freq=10, optimize_collision_model_freq=1,
                 random_field_points=10, collision_loss_koef=1, course_random_offset=1.5, collision_point_count=100,
                 rng="reference"):
        _lib.require_gpu()
        if not trajectory.is_cuda:
            raise _lib.NfoppError("the NFOPP hot path is HIP-only: create the planner with device='cuda' (got %s)"
                                  % trajectory.device)
        self._trajectory = trajectory
        self._device = trajectory.device
        self._collision_model = collision_model
        self._collision_checker = collision_checker
        self._collision_optimizer = collision_optimizer
        self._trajectory_optimizer = trajectory_optimizer
        self._random_sample_border = (0, 0, 0, 0)
        self._fine_random_offset = trajectory_random_offset
        self._collision_weight = collision_weight
        self._velocity_hessian_weight = velocity_hessian_weight
        self._init_collision_iteration = init_collision_iteration
        self._init_collision_points = init_collision_points
        self._reparametrize_trajectory_freq = reparametrize_trajectory_freq
        self._optimize_collision_model_freq = optimize_collision_model_freq
        self._random_field_points = random_field_points
        self._step_count = 0
        self._collision_loss_koef = collision_loss_koef
        self._previous_trajectory = None
        self._collision_positions = np.zeros((0, self.point_dim))
        self._collision_positions_times = np.zeros(0)
        self._course_random_offset = course_random_offset
        self._collision_point_count = collision_point_count
        self.checked_positions = np.zeros((0, 3))
        self.truth_collision = np.zeros(0, dtype=bool)
        self._rng = rng
        if rng not in ("reference", "device"):
            raise ValueError("rng must be 'reference' or 'device'")
        n, d = trajectory.shape
        if d != self.point_dim:
            raise ValueError("trajectory must be [N, %d]" % self.point_dim)
        self._engine = TrajectoryEngine(collision_model, 1, n, d, self._make_hyper(), velocity_hessian_weight,
                                        self._device, traj=trajectory.detach())
        self._inv_hessian = torch.tensor(self._engine.hinv, device=self._device)
        # ONF Adam state (flat, same order as the parameter buffer)
        self._onf_m = torch.zeros_like(collision_model.flat_parameters)
        self._onf_v = torch.zeros_like(collision_model.flat_parameters)
        self._onf_step = 0
        self._onf_grad = torch.zeros(collision_model.n_params + 2, dtype=torch.float32, device=self._device)
        self._onf_ws = None
        self.last_onf_loss = None

    # ---- configuration ---------------------------------------------------------------------------------------------
    def _make_hyper(self):
        lr, betas, eps = _adam_group(self._trajectory_optimizer)
        return TrajectoryHyper(collision_weight=self._collision_weight, lr=lr, betas=betas, eps=eps,
                               bounds=self._random_sample_border)

    def _sync_hyper(self):
        """Attributes may be edited between steps (drivers do): the kernel scalars follow them, rebuilt only when one changed."""
        h = self._make_hyper()
        key = (h.collision_weight, h.angle_weight, h.constraint_deltas_weight, h.multipliers_lr, h.collision_multipliers_lr,
               h.boundary_weight, h.collision_beta, h.direction_delta_weight, h.lr, h.betas, h.eps, h.bounds)
        if key != getattr(self, "_hyper_key", None):
            self._hyper_key = key
            self._engine.hyper = h

    @property
    def _start_point(self):
        return self._engine.start

    @property
    def _goal_point(self):
        return self._engine.goal

    # ---- step ---------------------------------------------------------------------------------------------------------
    def step(self, n=1):
        """The reference's `step()` (nerf:60-71).  `step(n)` runs n of them; the stretches between two ONF fitting steps
        (all n when `optimize_collision_model_freq` exceeds the horizon, i.e. a frozen field) are enqueued by ONE library
        call each (nfopp_traj_steps) instead of a Python round trip per step -- what a budgeted caller loop wants
        (nfop/ros/goal_planner_adapter.py:50-52: `while time < timeout: planner.step()` becomes `planner.step(chunk)`).
        Results are bit-identical to n single calls: the per-step draws are taken from the same generator in the same
        order.  On return the steps are enqueued; `get_path()` synchronises, as before."""
        n = int(n)
        while n > 0:
            freq = self._optimize_collision_model_freq
            if self._step_count % freq == 0:
                self._optimize_collision_model()
            run = min(n, freq - self._step_count % freq)      # steps until the next fit falls due
            if run > 1:
                self._sync_hyper()
                draws = [self._draw_t() for _ in range(run)]
                t_steps = None if draws[0] is None else torch.stack([d.reshape(-1) for d in draws])[:, None, :]
                with self._collision_model.frozen():           # the field cannot change inside the run
                    self._engine.steps(run, self._step_count, self._reparametrize_trajectory_freq, t_steps=t_steps,
                                       want_terms=True)
                self._step_count += run
            else:
                self._optimize_trajectory()
                if self._step_count % self._reparametrize_trajectory_freq == 0:
                    self.reparametrize_trajectory()
                self._step_count += 1
            n -= run

    def full_trajectory(self):
        return self._engine.full_trajectory()[0]

    def get_path(self):
        return self.full_trajectory().detach().cpu().numpy()

    # ---- ONF fitting (nerf:76-141) ---------------------------------------------------------------------------------
    def _optimize_collision_model(self, positions=None):
        if positions is None:
            if self._previous_trajectory is None:
                self._previous_trajectory = self._trajectory.detach().cpu().numpy().copy()
            positions = self._host_training_poses(self._previous_trajectory)
            self._previous_trajectory = self._trajectory.detach().cpu().numpy().copy()
        truth = self._calculate_truth_collision(positions)
        self._fit_step(positions, np.asarray(truth))

    def _fit_step(self, positions, truth):
        """One BCE/Adam step of the field on host-provided samples: gradient kernel + flat Adam kernel."""
        lib = _lib.load()
        model = self._collision_model
        # one upload for poses and labels (two pageable copies cost 2 x 25 us at B = 1): [P * D | P] floats, two contiguous views
        pos32 = np.ascontiguousarray(positions, dtype=np.float32)
        p, d = pos32.shape
        packed = torch.from_numpy(np.concatenate([pos32.reshape(-1), np.asarray(truth).astype(np.float32).reshape(-1)])).to(self._device)
        samples, labels = packed[:p * d].view(p, d), packed[p * d:]
        cfg = model.config_c()
        need = lib.nfopp_onf_train_workspace_bytes(cfg, p)
        if self._onf_ws is None or self._onf_ws.numel() * 4 < need:
            self._onf_ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=self._device)
        _lib.check(lib.nfopp_onf_train_grad(cfg, _lib.ptr(model.flat_parameters), _lib.ptr(samples), _lib.ptr(labels),
                                            p, 1.0 / p, _lib.ptr(self._onf_grad), _lib.ptr(self._onf_ws),
                                            self._onf_ws.numel() * 4, _lib.stream_ptr()))
        lr, (b1, b2), eps = _adam_group(self._collision_optimizer)
        self._onf_step += 1
        bc1 = 1 - b1 ** self._onf_step
        bc2 = 1 - b2 ** self._onf_step
        _lib.check(lib.nfopp_adam_step(_lib.ptr(model.flat_parameters), _lib.ptr(self._onf_grad), _lib.ptr(self._onf_m),
                                       _lib.ptr(self._onf_v), model.n_params, b2, 1 - b1, 1 - b2, eps, lr / bc1,
                                       bc2 ** 0.5, _lib.stream_ptr()))
        model.mark_modified()   # a raw-pointer write: torch's version counter does not see it
        self.last_onf_loss = self._onf_grad[model.n_params]

    def _calculate_truth_collision(self, positions):
        self.checked_positions = positions.copy()
        self.truth_collision = self._collision_checker.check_collision(positions)
        return self.truth_collision

    def _calculate_predicted_collision(self, positions):
        return self._collision_model(torch.tensor(np.ascontiguousarray(positions, dtype=np.float32), device=self._device))

    def _host_training_poses(self, previous):
        """Training poses of one fitting step for the B = 1 drop-in planner, formed on the host because the caller's
        collision checker is a host object (nerf:101-141; SE(2) variants constrained:49-61,173-176).  The fixtures g9 / g15 /
        g18 replay the reference's global numpy stream, so the ORDER and SHAPES of the draws are part of the contract:
            1. rand(N-1)                       where on each segment of the previous trajectory a pose is taken
            2. randn(N-1, 2) [+ randn(N-1)]    "course" copies: xy noise (and heading noise for SE(2))
            3. randn(N-1, 2) [+ randn(N-1)]    "fine" copies, appended to the retained pool
            4. choice(len, pool size, p=w)     retained-pool resampling -- only once the candidates outnumber the pool
            5. rand(F, 2) [+ rand(F, 1)]       uniform field poses (and headings)
        Returns course | retained pool | field poses; the pool and its ages are kept for the next step."""
        seg = np.random.rand(previous.shape[0] - 1).astype(np.float32)[:, None]                              # 1
        between = previous[1:] * (np.float32(1) - seg) + previous[:-1] * seg        # plain fp32 lerp, nerf:113-117
        course = self._jitter(between, self._course_random_offset)                                          # 2
        fine = self._jitter(between, self._fine_random_offset)                                              # 3
        candidates = np.concatenate([self._collision_positions, fine], axis=0)
        ages = np.concatenate([self._collision_positions_times, np.zeros(len(fine))], axis=0)
        keep = self._collision_point_count
        if len(candidates) >= keep:
            # weights sigmoid(logit) * exp(-0.03 age) + 1e-6 (nerf:122-133); sigmoid in fp32 like torch's
            logits = self._calculate_predicted_collision(candidates).detach().cpu().numpy()[:, 0]
            w = (np.float32(1) / (np.float32(1) + np.exp(-logits, dtype=np.float32))).astype(np.float32)
            w = w * np.exp(-ages * 0.03) + 1e-6
            w = w / np.sum(w)
            with_replacement = np.count_nonzero(w > 1e-6) < keep
            chosen = np.random.choice(len(candidates), keep, replace=with_replacement, p=w)                 # 4
            candidates, ages = candidates[chosen], (ages + 1)[chosen]
        self._collision_positions, self._collision_positions_times = candidates, ages
        return np.concatenate([course, candidates, self._uniform_poses(self._random_field_points)], axis=0)  # 5

    def _jitter(self, poses, xy_sigma):
        """Gaussian copies of poses.  2-D: a new float64 array (nerf:119-120); SE(2): an fp32 copy written in place, headings
        with `angle_offset` (constrained:57-61) -- the dtypes differ in the reference and both are kept."""
        if self.point_dim == 2:
            return poses + np.random.randn(poses.shape[0], 2) * xy_sigma
        out = poses.copy()
        out[:, :2] = out[:, :2] + np.random.randn(out.shape[0], 2) * xy_sigma
        out[:, 2] = out[:, 2] + np.random.randn(out.shape[0]) * self._angle_offset
        return out

    def _uniform_poses(self, count):
        """`count` poses uniform over the sampling border (nerf:135-141), SE(2): + a uniform heading (constrained:173-176)."""
        lo_x, hi_x, lo_y, hi_y = self._random_sample_border
        xy = np.random.rand(count, 2)
        xy[:, 0] = lo_x + xy[:, 0] * (hi_x - lo_x)
        xy[:, 1] = lo_y + xy[:, 1] * (hi_y - lo_y)
        if self.point_dim == 2:
            return xy
        return np.concatenate([xy, np.random.rand(count, 1) * 2 * np.pi], axis=1)

    # ---- trajectory optimisation (nerf:143-169) ------------------------------------------------------------------
    def _draw_t(self):
        if self._rng == "device":
            return None
        return torch.tensor(np.random.rand(self._trajectory.shape[0] - 1).astype(np.float32))

    def _optimize_trajectory(self):
        self._sync_hyper()
        self._engine.optimize_trajectory(self._draw_t())

    def trajectory_loss_terms(self):
        """Loss terms of the most recent trajectory step (dict of floats; synchronises)."""
        return {k: float(v[0]) for k, v in self._engine.loss_terms().items()}

    # ---- lifecycle (nerf:181-222) -------------------------------------------------------------------------------------
    def init(self, start_point, goal_point, boundaries):
        self._engine.set_endpoints(np.asarray(start_point, np.float32)[None], np.asarray(goal_point, np.float32)[None])
        self._random_sample_border = boundaries
        self._init_trajectory()
        self._init_collision_model()
        self._step_count = 0

    def _init_trajectory(self):
        n = self._trajectory.shape[0] + 2
        s, g = self._engine.start.cpu(), self._engine.goal.cpu()
        tr = torch.stack([torch.linspace(s[0, k], g[0, k], n)[1:-1] for k in range(2)], dim=1)
        self._engine.traj.copy_(tr)

    def _init_collision_model(self):
        for _ in range(self._init_collision_iteration):
            positions = self._uniform_poses(self._init_collision_points)
            self._optimize_collision_model(positions)

    def _endpoint_update(self, point, is_goal):
        eng = self._engine
        tr = eng.traj.view(eng.N, eng.D)
        (eng.goal if is_goal else eng.start).copy_(torch.tensor(np.asarray(point, np.float32))[None])
        ref = eng.goal if is_goal else eng.start
        return tr, ref

    def update_goal_point(self, goal_point):
        tr, ref = self._endpoint_update(goal_point, True)
        min_index = int(torch.argmin(torch.sum((tr - ref) ** 2, dim=1)))
        tr[min_index:] = ref
        self.reparametrize_trajectory()
        self._step_count = 0

    def update_start_point(self, start_point):
        tr, ref = self._endpoint_update(start_point, False)
        min_index = int(torch.argmin(torch.sum((tr - ref) ** 2, dim=1)))
        tr[:min_index] = ref
        self.reparametrize_trajectory()
        self._step_count = 0

    def set_boundaries(self, boundaries):
        self._random_sample_border = boundaries
        self._step_count = 0

    def reparametrize_trajectory(self):
        self._engine.reparametrize()


class ConstrainedNERFOptPlanner(NERFOptPlanner):
    """SE(2) planner (nfop/constrained_nerf_opt_planner.py:12-194)."""

    point_dim = 3

    def __init__(self, trajectory, collision_model, collision_checker, collision_optimizer, trajectory_optimizer,
                 trajectory_initializer, trajectory_random_offset, collision_weight, velocity_hessian_weight,
                 init_collision_iteration=100, init_collision_points=100, reparametrize_trajectory_freq=10,
                 optimize_collision_model_freq=1, random_field_points=10, angle_weight=0.5, constraint_deltas_weight=20,
                 multipliers_lr=1e-1, boundary_weight=1, collision_multipliers_lr=1e-3, angle_offset=0,
                 collision_beta=1, direction_delta_weight=0, rng="reference"):
        self._angle_weight = angle_weight
        self._constraint_delta_weight = constraint_deltas_weight
        self._multipliers_lr = multipliers_lr
        self._collision_multipliers_lr = collision_multipliers_lr
        self._boundary_weight = boundary_weight
        self._angle_offset = angle_offset
        self._collision_beta = collision_beta
        self._direction_delta_weight = direction_delta_weight
        self._trajectory_initializer = trajectory_initializer
        # like the reference, course_random_offset / collision_point_count / collision_loss_koef are NOT forwarded
        super().__init__(trajectory, collision_model, collision_checker, collision_optimizer, trajectory_optimizer,
                         trajectory_random_offset, collision_weight, velocity_hessian_weight, init_collision_iteration,
                         init_collision_points, reparametrize_trajectory_freq, optimize_collision_model_freq,
                         random_field_points, rng=rng)

    def _make_hyper(self):
        lr, betas, eps = _adam_group(self._trajectory_optimizer)
        return TrajectoryHyper(self._collision_weight, self._angle_weight, self._constraint_delta_weight,
                               self._multipliers_lr, self._collision_multipliers_lr, self._boundary_weight,
                               self._collision_beta, self._direction_delta_weight, lr, betas, eps,
                               self._random_sample_border)

    @property
    def _constraint_multipliers(self):
        return self._engine.lam[0]

    @property
    def _collision_multipliers(self):
        return self._engine.cm[0]

    def _init_trajectory(self):
        ini = self._trajectory_initializer
        if type(ini) is TrajectoryInitializer:
            # the stock initialiser runs as one device kernel (csrc/traj_init.hip); any other object the caller passes
            # (e.g. an A* seeder) keeps the reference's host protocol below
            init_trajectories(self._engine.start, self._engine.goal, self._engine.N, ini._init_angles_with_trajectory,
                              out=self._engine.traj)
            return
        tr = torch.zeros(self._trajectory.shape[0], 3)
        ini.initialize_trajectory(tr, self._engine.start.cpu(), self._engine.goal.cpu())
        self._engine.traj.copy_(tr)

    def _calculate_truth_collision(self, positions):
        self.checked_positions = Position2.from_vec(positions)
        self.truth_collision = self._collision_checker.check_collision(self.checked_positions)
        return self.truth_collision

    def _draw_t(self):
        if self._rng == "device":
            return None
        return torch.rand(self._trajectory.shape[0] - 1, 1)[:, 0]

    def _endpoint_min_index(self, tr, ref):
        delta = torch.sum((tr[:, :2] - ref[:, :2]) ** 2, dim=1)
        return min(int(torch.argmin(delta)) + 1, tr.shape[0])

    def update_goal_point(self, goal_point):
        tr, ref = self._endpoint_update(goal_point, True)
        tr[self._endpoint_min_index(tr, ref):] = ref
        self.reparametrize_trajectory()
        self._step_count = 0

    def update_start_point(self, start_point):
        tr, ref = self._endpoint_update(start_point, False)
        tr[:self._endpoint_min_index(tr, ref)] = ref
        self.reparametrize_trajectory()
        self._step_count = 0
