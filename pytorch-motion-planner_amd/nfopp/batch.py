"""Batched / sharded planning over B independent trajectories that share one ONF (the batch axis is new: the
reference plans one trajectory per process, nfop/planner_factory.py:55,69).

* `BatchPlanner`   -- B trajectories on one GPU: init / step / get_paths, frozen or continuously fitted ONF.
* `shard_range`    -- contiguous split of a global batch over ranks; each rank passes its first global index as
                      `traj_index_offset` so the in-kernel Philox stream (and hence every result) is independent of
                      how the batch is sharded.  The frozen-ONF step needs NO collective.
* `OnfFitter`      -- ONF fitting step for data-parallel continuous learning: local gradient kernel (normalised by
                      the GLOBAL sample count), one all-reduce(SUM) of the flat [n_params + 2] buffer (RCCL over
                      xGMI when the process group is "nccl"), then the identical Adam step on every rank, so the
                      replicated weights stay bit-identical across ranks.
"""
import numpy as np
import torch

from . import _lib
from .engine import TrajectoryEngine, TrajectoryHyper
from .path_tools import init_trajectories


def shard_range(global_batch, rank, world_size):
    """[lo, hi) of the trajectories owned by `rank` (contiguous, sizes differ by at most one)."""
    base, rem = divmod(int(global_batch), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _linspace_rows(a, b, steps):
    """Row-wise torch.linspace (fp32 CPU rounding: fp32 step, one fused multiply-add per element, two-sided)."""
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    step = ((b - a) / np.float32(steps - 1)).astype(np.float32).astype(np.float64)
    i = np.arange(steps)
    lo = (a.astype(np.float64)[:, None] + step[:, None] * i[None]).astype(np.float32)
    hi = (b.astype(np.float64)[:, None] - step[:, None] * (steps - 1 - i)[None]).astype(np.float32)
    return np.where(i[None] < steps // 2, lo, hi)


def straight_line_init(starts, goals, n_waypoints):
    """Batched TrajectoryInitializer (nfop/trajectory_initializer.py:12-29): xy on the segment start->goal, theta
    interpolated along the wrapped shortest rotation.  Host numpy, one-time per `init`."""
    starts = np.asarray(starts, np.float32)
    goals = np.asarray(goals, np.float32)
    d = starts.shape[1]
    out = np.zeros((starts.shape[0], n_waypoints, d), np.float32)
    for k in range(2):
        out[:, :, k] = _linspace_rows(starts[:, k], goals[:, k], n_waypoints + 2)[:, 1:-1]
    if d == 3:
        pi, two_pi = np.float32(np.pi), np.float32(2 * np.pi)
        delta = (np.remainder((goals[:, 2] - starts[:, 2]) + pi, two_pi).astype(np.float32) - pi).astype(np.float32)
        out[:, :, 2] = _linspace_rows(starts[:, 2], (delta + starts[:, 2]).astype(np.float32), n_waypoints + 2)[:, 1:-1]
    return out


class OnfFitter(object):
    """One BCE/Adam step of the shared field on this rank's samples; gradients summed over `group` first."""

    def __init__(self, onf, lr, betas, eps=1e-8, group=None, grad_fn=None, distributed=True):
        self.onf, self.lr, self.betas, self.eps, self.group = onf, float(lr), tuple(betas), float(eps), group
        # distributed=False: purely local fit even inside an initialised process group (e.g. identical pre-fits)
        self.distributed = bool(distributed)
        flat = onf.flat_parameters
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self.step_count = 0
        self.grad = torch.zeros(onf.n_params + 2, dtype=torch.float32, device=flat.device)
        self._ws = None
        self._grad_fn = grad_fn or self._hip_grad
        self._adam_fn = self._hip_adam if grad_fn is None else None
        self.last_loss = None

    def _hip_grad(self, samples, labels, inv_count):
        lib = _lib.load()
        cfg = self.onf.config_c()
        p = samples.shape[0]
        need = lib.nfopp_onf_train_workspace_bytes(cfg, p)
        if self._ws is None or self._ws.numel() * 4 < need:
            self._ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=samples.device)
        _lib.check(lib.nfopp_onf_train_grad(cfg, _lib.ptr(self.onf.flat_parameters), _lib.ptr(samples),
                                            _lib.ptr(labels), p, inv_count, _lib.ptr(self.grad), _lib.ptr(self._ws),
                                            self._ws.numel() * 4, _lib.stream_ptr()))

    def _hip_adam(self, step_size, bc2_sqrt):
        b1, b2 = self.betas
        _lib.check(_lib.load().nfopp_adam_step(_lib.ptr(self.onf.flat_parameters), _lib.ptr(self.grad), _lib.ptr(self.m),
                                               _lib.ptr(self.v), self.onf.n_params, b2, 1 - b1, 1 - b2, self.eps,
                                               step_size, bc2_sqrt, _lib.stream_ptr()))
        self.onf.mark_modified()   # a raw-pointer write: torch's version counter does not see it

    def _in_group(self):
        return self.distributed and torch.distributed.is_available() and torch.distributed.is_initialized()

    def global_count(self, local_count):
        """Sample count over all ranks when the caller cannot state it: ONE extra all-reduce and a host sync.  The
        hot loop never takes this path -- `BatchPlanner` passes the count it knows statically (global batch x poses per
        trajectory); it exists for ragged, caller-managed sample sets."""
        if not self._in_group():
            return int(local_count)
        gloo = torch.distributed.get_backend(self.group) == "gloo"
        c = torch.tensor([float(local_count)], dtype=torch.float64, device="cpu" if gloo else self.grad.device)
        torch.distributed.all_reduce(c, group=self.group)
        return int(c.item())

    def world_size(self):
        return torch.distributed.get_world_size(self.group) if self._in_group() else 1

    def step(self, samples, labels, global_count=None, adam_fn=None):
        """samples [P_local, point_dim], labels [P_local] on this rank's device.  `global_count` = number of samples
        over ALL ranks (the BCE mean's denominator); pass it whenever it is known without communication.  Returns
        the global mean loss (a device scalar; no host sync)."""
        p = samples.shape[0]
        total = self.global_count(p) if global_count is None else int(global_count)
        self._grad_fn(samples, labels, 1.0 / total)
        if self._in_group():
            if self.grad.is_cuda and torch.distributed.get_backend(self.group) == "gloo":
                host = self.grad.cpu()           # rehearsal path only: gloo reduces host buffers
                torch.distributed.all_reduce(host, group=self.group)
                self.grad.copy_(host)
            else:
                torch.distributed.all_reduce(self.grad, group=self.group)   # SUM; RCCL on the "nccl" backend
        self.step_count += 1
        b1, b2 = self.betas
        step_size = self.lr / (1 - b1 ** self.step_count)
        bc2_sqrt = (1 - b2 ** self.step_count) ** 0.5
        (adam_fn or self._adam_fn)(step_size, bc2_sqrt)
        self.last_loss = self.grad[self.onf.n_params]
        return self.last_loss


class BatchPlanner(object):
    """B trajectories, one shared ONF, one GPU.  `step()` = the planner step of nfop/nerf_opt_planner.py:60-71 for the
    whole batch: [ONF fit on freshly sampled poses, when a ground-truth `checker` is given] -> one
    `_optimize_trajectory` per trajectory -> periodic reparametrisation.  Without a checker the field is frozen.

    The planner owns the field's update loop, so it FREEZES the ONF object (`ONF.freeze()`): launches reuse the pre-split
    weight image until the planner's own Adam step (or an in-place torch op) changes the parameters.  A caller who writes
    the parameters behind torch's back while a planner exists -- `dist.broadcast(onf.flat_parameters, 0)`, `.data`
    assignments -- must call `onf.mark_modified()` afterwards (or construct with `freeze_field=False`)."""

    def __init__(self, onf, batch, n_waypoints, hyper, velocity_hessian_weight=0.5, reparametrize_trajectory_freq=10,
                 device="cuda", seed=0, traj_index_offset=0, checker=None, fit_lr=2e-2, fit_betas=(0.9, 0.9),
                 optimize_collision_model_freq=1, trajectory_random_offset=0.02, course_random_offset=1.5,
                 angle_offset=0.0, random_field_points=10, collision_point_count=100, group=None,
                 init_angles_with_trajectory=False, global_batch=None, freeze_field=True):
        if freeze_field:
            onf.freeze()
        self.init_angles_with_trajectory = bool(init_angles_with_trajectory)
        # trajectories over ALL ranks (continuous learning: denominator of the BCE mean); default = equal shards
        self.global_batch = None if global_batch is None else int(global_batch)
        self.engine = TrajectoryEngine(onf, batch, n_waypoints, onf.point_dim, hyper, velocity_hessian_weight, device,
                                       seed=seed, traj_index_offset=traj_index_offset)
        self.onf = onf
        self.reparam_freq = int(reparametrize_trajectory_freq)
        self.step_count = 0
        self.checker = checker
        self.fit_freq = int(optimize_collision_model_freq)
        self.sampler = self.fitter = self._prev = None
        if checker is not None:
            from .learning import BatchSampler
            self.sampler = BatchSampler(onf, batch, n_waypoints, course_random_offset, trajectory_random_offset,
                                        angle_offset, random_field_points, collision_point_count, device,
                                        seed=seed + 1, traj_index_offset=traj_index_offset)
            self.fitter = OnfFitter(onf, fit_lr, fit_betas, group=group)

    def init(self, starts, goals, boundaries, trajectories=None):
        eng = self.engine
        eng.set_endpoints(starts, goals)
        h = eng.hyper
        eng.hyper = TrajectoryHyper(h.collision_weight, h.angle_weight, h.constraint_deltas_weight, h.multipliers_lr,
                                    h.collision_multipliers_lr, h.boundary_weight, h.collision_beta,
                                    h.direction_delta_weight, h.lr, h.betas, h.eps, boundaries)
        if trajectories is None:
            # device initialiser (trajectory_initializer.py:12-45); eng.start / eng.goal were uploaded just above
            init_trajectories(eng.start, eng.goal, eng.N, self.init_angles_with_trajectory and eng.D == 3, out=eng.traj)
        else:
            eng.traj.copy_(torch.as_tensor(np.asarray(trajectories, np.float32)).reshape(eng.traj.shape))
        for buf in (eng.lam, eng.cm, eng.adam_m, eng.adam_v):
            if buf is not None:
                buf.zero_()
        eng.adam_step = 0
        self.step_count = 0
        self._prev = None

    def fit_field(self):
        """One `_optimize_collision_model` over the batch (nerf:76-91): poses from the PREVIOUS trajectories, labels
        from the device checker, gradient all-reduced over the process group, identical Adam on every rank."""
        eng = self.engine
        if self._prev is None:
            self._prev = eng.traj.detach().clone()
        samples = self.sampler.draw(self._prev, eng.hyper.bounds)
        self._prev.copy_(eng.traj)
        labels = self.checker.labels(samples, out=self.sampler.labels)
        # the sample count over all ranks is known statically: no count all-reduce, no host sync in the step
        gb = self.global_batch if self.global_batch is not None else self.fitter.world_size() * eng.B
        return self.fitter.step(samples, labels, global_count=gb * self.sampler.S)

    def step(self, t=None, want_terms=False, n=1):
        """One planner step for the whole batch -- or `n` of them (`step(n=...)`): with a frozen field (no checker) the n
        steps are enqueued by ONE library call (nfopp_traj_steps), without returning to Python in between -- the form
        the callers' loops want (nfop/ros/goal_planner_adapter.py:50-52, scripts/run_planner.py:76-77).  With continuous
        learning every fit needs fresh samples, so the steps run one by one.  `t` [B, N-1] (n = 1) or [n, B, N-1]
        injects the draws.  Bit-identical to n calls of `step()`."""
        n = int(n)
        if n > 1 or (n == 1 and self.checker is None and t is None):
            if self.checker is None:
                self.engine.steps(n, self.step_count, self.reparam_freq, t_steps=t, want_terms=want_terms)
                self.step_count += n
                return
            for k in range(n):
                self.step(None if t is None else t[k], want_terms=want_terms and k == n - 1)
            return
        if n < 1:
            return
        if self.checker is not None and self.step_count % self.fit_freq == 0:
            self.fit_field()
        self.engine.optimize_trajectory(t, want_terms=want_terms)
        if self.step_count % self.reparam_freq == 0:
            self.engine.reparametrize()
        self.step_count += 1

    def get_paths(self):
        return self.engine.full_trajectory().detach().cpu().numpy()

    # ---- path evaluation, best-path bookkeeping, early stop (scripts/run_bench_mr.py:109-132 for the batch) ---------
    def evaluate(self, checker=None, sub=4, early_stop=False):
        """Densifies every path (`sub` poses per segment), labels the poses with the ground-truth `checker`, keeps
        the shortest collision-free path per trajectory and -- with early_stop -- retires trajectories that are
        collision-free but no longer improving.  Returns device tensors (collides uint8 [B], length [B])."""
        from . import _lib as L
        checker = checker or self.checker
        if checker is None:
            raise ValueError("evaluate() needs a ground-truth checker")
        eng = self.engine
        B, N, D = eng.B, eng.N, eng.D
        m = (N + 1) * int(sub) + 1
        f32 = dict(dtype=torch.float32, device=eng.device)
        if getattr(self, "_eval_sub", None) != sub:
            self._eval_sub = sub
            self._poses = torch.empty(B, m, D, **f32)
            self._pose_labels = torch.empty(B * m, **f32)
            self._length = torch.empty(B, **f32)
            self._collides = torch.zeros(B, dtype=torch.uint8, device=eng.device)
        if not hasattr(self, "best_length"):
            self.best_length = torch.full((B,), float("inf"), **f32)
            self.best_traj = eng.traj.detach().clone().view(B, N, D)
        if early_stop and eng.active is None:
            eng.active = torch.ones(B, dtype=torch.uint8, device=eng.device)
        lib = L.load()
        L.check(lib.nfopp_path_interpolate(L.ptr(eng.traj), L.ptr(eng.start), L.ptr(eng.goal), B, N, D, int(sub),
                                           L.ptr(self._poses), L.ptr(self._length), L.stream_ptr()))
        checker.labels(self._poses.view(B * m, D), out=self._pose_labels)
        L.check(lib.nfopp_path_select_best(L.ptr(self._pose_labels), L.ptr(self._length), L.ptr(eng.traj), B, m, N, D,
                                           L.ptr(self.best_traj), L.ptr(self.best_length),
                                           L.ptr(self._collides, torch.uint8),
                                           L.ptr(eng.active, torch.uint8) if early_stop else None, L.stream_ptr()))
        return self._collides, self._length

    def best_paths(self):
        """[B, N+2, D]: the best collision-free path found so far, the current path where none was found yet."""
        eng = self.engine
        if not hasattr(self, "best_length"):
            return self.get_paths()
        found = torch.isfinite(self.best_length)[:, None, None]
        tr = torch.where(found, self.best_traj, eng.traj.view(eng.B, eng.N, eng.D))
        return torch.cat([eng.start[:, None], tr, eng.goal[:, None]], dim=1).cpu().numpy()
