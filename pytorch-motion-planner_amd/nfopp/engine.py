"""Batched trajectory-optimisation engine: device state + the HIP pipeline of one planner step.

State of B independent trajectories (the reference holds exactly one, nfop/nerf_opt_planner.py:16-43,
nfop/constrained_nerf_opt_planner.py:23-40) lives in HBM as contiguous fp32 arrays:
    traj [B,N,D]   start/goal [B,D]   lam [B,N+1]   cm [B,N]   adam_m/adam_v [B,N,D]
plus per-step scratch  t [B,N-1]  and  onf_out [B,N-1,4].
One step = 2 launches:  nfopp_traj_collision_eval (fused sampling + ONF fwd/bwd, MFMA)  ->  nfopp_traj_update
(stencil terms, banded H^-1, Adam, multiplier ascent); every `reparam_freq` steps a third: nfopp_reparametrize.
"""
import math

import numpy as np
import torch

from . import _lib


def inverse_hessian(n, weight):
    """nfop/nerf_opt_planner.py:45-58: float64 inverse of weight * tridiag(-2, 4, -2) + I, rounded to fp32."""
    k = np.zeros((n, n), np.float32)
    idx = np.arange(n)
    k[idx, idx] = 4
    k[idx[1:], idx[:-1]] = -2
    k[idx[:-1], idx[1:]] = -2
    return np.linalg.inv(weight * k + np.eye(n)).astype(np.float32)


def band_of(hinv, rel_tol=1e-9):
    """Transposed band [2W+1, N] of a (symmetric, exponentially decaying) dense inverse Hessian.

    W is the smallest half-width such that every dropped entry is below rel_tol * max|entry| -- far below one
    fp32 ulp of any row sum, so the banded product equals the reference's dense `inv_hessian @ grad` (nerf:151)
    to rounding.  band[k, i] = hinv[i, i + k - W]."""
    n = hinv.shape[0]
    thr = float(np.abs(hinv).max()) * rel_tol
    w = 0
    for k in range(n - 1, 0, -1):
        if max(np.abs(np.diagonal(hinv, k)).max(), np.abs(np.diagonal(hinv, -k)).max()) >= thr:
            w = k
            break
    band = np.zeros((2 * w + 1, n), np.float32)
    for k in range(2 * w + 1):
        off = k - w
        d = np.diagonal(hinv, off)
        if off >= 0:
            band[k, :n - off] = d
        else:
            band[k, -off:] = d
    return band, w


def interior_range(band):
    """[lo, hi): the longest run of waypoints around the centre whose band column is bit-identical to the centre's
    (the inverse of a tridiagonal Toeplitz matrix is Toeplitz away from the ends: the boundary terms decay like
    r^(2*distance) and vanish below fp32 resolution).  The kernel broadcasts these coefficients from LDS."""
    n = band.shape[1]
    c = n // 2
    same = np.all(band == band[:, c:c + 1], axis=0)
    lo, hi = c, c + 1
    while lo > 0 and same[lo - 1]:
        lo -= 1
    while hi < n and same[hi]:
        hi += 1
    return (lo, hi) if hi - lo >= 64 else (0, 0)


class TrajectoryHyper(object):
    """Scalars of the trajectory step (nfop/constrained_nerf_opt_planner.py:13-40 + the Adam group)."""

    def __init__(self, collision_weight=1.0, angle_weight=0.5, constraint_deltas_weight=20.0, multipliers_lr=0.1,
                 collision_multipliers_lr=1e-3, boundary_weight=1.0, collision_beta=1.0, direction_delta_weight=0.0,
                 lr=1e-2, betas=(0.9, 0.9), eps=1e-8, bounds=(0.0, 0.0, 0.0, 0.0)):
        self.collision_weight = collision_weight
        self.angle_weight = angle_weight
        self.constraint_deltas_weight = constraint_deltas_weight
        self.multipliers_lr = multipliers_lr
        self.collision_multipliers_lr = collision_multipliers_lr
        self.boundary_weight = boundary_weight
        self.collision_beta = collision_beta
        self.direction_delta_weight = direction_delta_weight
        self.lr, self.betas, self.eps = lr, tuple(betas), eps
        self.bounds = tuple(bounds)

    def to_c(self, adam_step):
        """`adam_step` = 1-based count of the step being taken.  Scalars are formed in Python doubles like torch."""
        b1, b2 = self.betas
        bc1 = 1 - b1 ** adam_step
        bc2 = 1 - b2 ** adam_step
        c = getattr(self, "_c_block", None)
        if c is not None:                    # the step-invariant fields were filled when the block was made (the object is
            c.adam_step_size = self.lr / bc1   # treated as immutable: the planners build a new one when a scalar changes)
            c.adam_bc2_sqrt = math.sqrt(bc2)
            return c
        c = _lib.TrajHyperC()
        c.collision_weight = self.collision_weight
        c.angle_weight = self.angle_weight
        c.constraint_deltas_weight = self.constraint_deltas_weight
        c.multipliers_lr = self.multipliers_lr
        c.collision_multipliers_lr = self.collision_multipliers_lr
        c.boundary_weight = self.boundary_weight
        c.collision_beta = self.collision_beta
        c.direction_delta_weight = self.direction_delta_weight
        for k in range(4):
            c.bounds[k] = self.bounds[k]
        c.adam_beta2 = b2
        c.adam_omb1 = 1 - b1
        c.adam_omb2 = 1 - b2
        c.adam_eps = self.eps
        c.adam_step_size = self.lr / bc1
        c.adam_bc2_sqrt = math.sqrt(bc2)
        self._c_block = c
        return c


class TrajectoryEngine(object):
    """Owns the batched state and runs the step pipeline.  `traj` may be an externally owned [B,N,D] (or [N,D])
    HIP tensor (the planner's `_trajectory`); it is updated in place."""

    def __init__(self, onf, batch, n_waypoints, dim, hyper, velocity_hessian_weight, device, traj=None, seed=0,
                 traj_index_offset=0):
        _lib.require_gpu()
        self.onf, self.hyper = onf, hyper
        self.B, self.N, self.D = int(batch), int(n_waypoints), int(dim)
        if self.D != onf.point_dim:
            raise ValueError("trajectory dim %d does not match the ONF point dim %d" % (self.D, onf.point_dim))
        self.device = torch.device(device)
        f32 = dict(dtype=torch.float32, device=self.device)
        B, N, D = self.B, self.N, self.D
        if traj is None:
            traj = torch.zeros(B, N, D, **f32)
        self.traj = traj
        self.start = torch.zeros(B, D, **f32)
        self.goal = torch.zeros(B, D, **f32)
        self.lam = torch.zeros(B, N + 1, **f32) if D == 3 else None
        self.cm = torch.zeros(B, N, **f32) if D == 3 else None
        self.adam_m = torch.zeros(B, N, D, **f32)
        self.adam_v = torch.zeros(B, N, D, **f32)
        self.adam_step = 0
        self.t = torch.zeros(B, N - 1, **f32)
        self.onf_out = torch.zeros(B, N - 1, 4, **f32)
        self.terms = torch.zeros(B, _lib.NUM_TERMS, **f32)
        self.hinv = inverse_hessian(N, velocity_hessian_weight)
        band, self.half_width = band_of(self.hinv)
        self.interior = interior_range(band)
        self.hinv_band = torch.tensor(band, **f32)
        self.u = torch.linspace(0, 1, N + 2)[1:-1].contiguous().to(self.device)  # CPU linspace: reference rounding
        self.seed, self.rng_offset, self.traj_index_offset = int(seed), 0, int(traj_index_offset)
        # optional uint8 [B] mask: 0 = retired trajectory (early stop).  Retired trajectories are compacted out of the
        # ONF kernel's sample stream and skipped by update / reparam: their state stays bit for bit as it was.
        self.active = None
        self._live = None    # int32 [B+1] live-list workspace of nfopp_traj_collision_eval
        self._check_traj()

    def _check_traj(self):
        t = self.traj
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == self.B * self.N * self.D):
            raise _lib.NfoppError("trajectory must be a contiguous fp32 HIP tensor with %d elements" %
                                  (self.B * self.N * self.D))

    # ---- pipeline stages ------------------------------------------------------------------------------------------
    def collision_eval(self, t=None):
        """ONF logits + input gradients at the collision samples.  `t` [B,N-1] injects the draws (parity mode);
        None draws them on the device with Philox (counter = global sample index, rng_offset)."""
        lib = _lib.load()
        if t is not None:
            self.t.copy_(torch.as_tensor(t, dtype=torch.float32).reshape(self.B, self.N - 1))
            mode = 0
        else:
            mode = 1
        cfg = self.onf.config_c()
        if self.active is not None and self._live is None:
            self._live = torch.zeros(self.B + 1, dtype=torch.int32, device=self.device)
        _lib.check(lib.nfopp_traj_collision_eval(cfg, _lib.ptr(self.onf.flat_parameters), _lib.ptr(self.traj), self.B,
                                                 self.N, self.D, _lib.ptr(self.t), mode, self.seed, self.rng_offset,
                                                 self.traj_index_offset, _lib.ptr(self.onf_out),
                                                 _lib.ptr(self.active, torch.uint8),
                                                 _lib.ptr(self._live, torch.int32) if self.active is not None else None,
                                                 _lib.stream_ptr()))
        if mode == 1:
            self.rng_offset += 1

    def update(self, want_terms=True):
        lib = _lib.load()
        self.adam_step += 1
        hp = self.hyper.to_c(self.adam_step)
        _lib.check(lib.nfopp_traj_update(hp, self.B, self.N, self.D, _lib.ptr(self.traj), _lib.ptr(self.start),
                                         _lib.ptr(self.goal), _lib.ptr(self.lam), _lib.ptr(self.cm),
                                         _lib.ptr(self.adam_m), _lib.ptr(self.adam_v), _lib.ptr(self.t),
                                         _lib.ptr(self.onf_out), _lib.ptr(self.hinv_band), self.half_width,
                                         self.interior[0], self.interior[1],
                                         _lib.ptr(self.terms) if want_terms else None,
                                         _lib.ptr(self.active, torch.uint8), _lib.stream_ptr()))

    def optimize_trajectory(self, t=None, want_terms=True):
        """One `_optimize_trajectory` (nfop/nerf_opt_planner.py:143-155 + constrained:63-74) for the whole batch."""
        self.collision_eval(t)
        self.update(want_terms)

    def steps(self, n, step_count, reparam_freq, t_steps=None, want_terms=False):
        """`n` frozen-field planner steps enqueued by ONE library call (nfopp_traj_steps, ABI 6): per step
        collision_eval -> update -> reparametrize when (step_count + k) % reparam_freq == 0, with each step's Adam scalars
        formed on the C side as `TrajectoryHyper.to_c` forms them.  Bit-identical to n single steps.  `t_steps`
        [n, B, N-1] injects the draws (parity mode), None = in-kernel Philox (word rng_offset + k).  The ONF must not
        change during the call; nothing synchronises."""
        n = int(n)
        if n <= 0:
            return
        lib = _lib.load()
        cfg = self.onf.config_c()
        hp = self.hyper.to_c(self.adam_step + 1)
        if self.active is not None and self._live is None:
            self._live = torch.zeros(self.B + 1, dtype=torch.int32, device=self.device)
        P = _lib.ptr
        buf = _lib.TrajBuffersC(P(self.traj), P(self.start), P(self.goal), P(self.lam), P(self.cm), P(self.adam_m),
                                P(self.adam_v), P(self.t), P(self.onf_out), P(self.hinv_band), P(self.u),
                                P(self.active, torch.uint8), P(self._live, torch.int32) if self.active is not None else None,
                                self.B, self.N, self.D, self.half_width, self.interior[0], self.interior[1])
        tdev = None
        if t_steps is not None:
            tdev = torch.as_tensor(t_steps, dtype=torch.float32).reshape(n, self.B, self.N - 1).to(self.device).contiguous()
        b1, b2 = self.hyper.betas
        sched = _lib.StepScheduleC(float(self.hyper.lr), float(b1), float(b2), self.adam_step, int(step_count),
                                   self.traj_index_offset, self.seed, self.rng_offset, int(reparam_freq),
                                   0 if tdev is not None else 1)
        _lib.check(lib.nfopp_traj_steps(cfg, P(self.onf.flat_parameters), hp, buf, sched, n, P(tdev),
                                        P(self.terms) if want_terms else None, _lib.stream_ptr()))
        self.adam_step += n
        if tdev is None:
            self.rng_offset += n
        else:
            self._t_steps_keepalive = tdev   # the kernels read it asynchronously; the last step's row is also K2's `t`

    def reparametrize(self):
        lib = _lib.load()
        _lib.check(lib.nfopp_reparametrize(self.B, self.N, self.D, _lib.ptr(self.traj), _lib.ptr(self.start),
                                           _lib.ptr(self.goal), _lib.ptr(self.lam), _lib.ptr(self.cm),
                                           _lib.ptr(self.u), _lib.ptr(self.active, torch.uint8), _lib.stream_ptr()))

    # ---- helpers --------------------------------------------------------------------------------------------------
    def set_endpoints(self, start, goal):
        self.start.copy_(torch.as_tensor(np.asarray(start, np.float32)).reshape(self.B, self.D))
        self.goal.copy_(torch.as_tensor(np.asarray(goal, np.float32)).reshape(self.B, self.D))

    def full_trajectory(self):
        tr = self.traj.view(self.B, self.N, self.D)
        return torch.cat([self.start[:, None], tr, self.goal[:, None]], dim=1)

    def loss_terms(self):
        """Per-trajectory loss terms of the last update as a dict of [B] numpy arrays (synchronises)."""
        t = self.terms.cpu().numpy()
        return {name: t[:, k] for k, name in enumerate(_lib.TERM_NAMES)}
