"""CPU baselines for bench.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rule as oracle/nfopp_oracle.py: only
tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import or run this file).

BASELINE.md section 3 / SURVEY 8(d) ask for two CPU figures next to the GPU number, because the reference's Python never
travels to the GPU box:

  (1) `BatchedTorchPlanner`  -- the STRONG baseline: the planner step restated for a batch, autograd-free (closed-form
      gradients of SURVEY Appendix A), every dense layer one MKL sgemm over B*(N-1) rows, all host cores
      (`torch.set_num_threads`).  This is what a CPU user would write after reading the GPU design.
  (2) `EagerAutogradPlanner` -- the REFERENCE-FAITHFUL baseline: ONE trajectory, eager PyTorch ops + autograd +
      `torch.optim.Adam`, the op sequence of the reference's hot path restated op for op
      (nfop/constrained_nerf_opt_planner.py:63-130, nfop/nerf_opt_planner.py:143-176, nfop/onf_model.py:33-50,
      nfop/angle_encoder.py:15-18, frozen field).  It is dispatch-bound exactly like the reference (the true
      reference measured 3.9 ms/step at N = 256 in the build container); timed on one core and process-parallel.

Both are validated against the reference-generated fixtures in tests/test_cpu_baselines.py.  Run as a script
(`python oracle/cpu_baselines.py --input x.npz ...`) it times both on a bounded sample and prints one JSON line; bench.py
starts it as a child process that never touches the GPU.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

HIDDEN = 100
PI = math.pi


def wrap(a):
    """nfop/torch_math.py:5-7"""
    return torch.remainder(a + PI, 2 * PI) - PI


class Field(object):
    """Frozen ONF weights as plain tensors, state_dict order of the flat buffer."""

    def __init__(self, flat, mean, sigma, use_cos=True, bias=True, angle_dim=10):
        flat = torch.as_tensor(np.asarray(flat, np.float32))
        self.mean, self.sigma, self.use_cos, self.angle_dim = float(mean), float(sigma), bool(use_cos), int(angle_dim)
        n_enc = 200 if use_cos else 100
        f = n_enc + 2 * angle_dim
        o = 0

        def take(*shape):
            nonlocal o
            n = int(np.prod(shape))
            v = flat[o:o + n].reshape(shape).clone()
            o += n
            return v
        if angle_dim:
            self.ang_b, self.ang_f = take(2 * angle_dim), take(2 * angle_dim)
        self.w1, self.b1 = take(HIDDEN, f), take(HIDDEN)
        self.w2, self.b2 = take(HIDDEN, HIDDEN), take(HIDDEN)
        self.w3, self.b3 = take(1, HIDDEN + f), take(1)
        self.we = take(n_enc, 2)
        self.be = take(n_enc) if bias else torch.zeros(n_enc)
        assert o == flat.numel(), (o, flat.numel())
        self.w1t, self.w2t, self.wet = self.w1.t().contiguous(), self.w2.t().contiguous(), self.we.t().contiguous()

    # ---- autograd path (eager baseline): plain differentiable ops, as ONF.forward composes them -------------------
    def logits(self, x):
        u = (x[:, :2] - self.mean) / self.sigma
        e = torch.nn.functional.linear(u, self.we, self.be)
        feats = torch.cat([torch.sin(e[:, :100]), torch.cos(e[:, 100:])], dim=1) if self.use_cos else torch.sin(e)
        if self.angle_dim:
            z = (x[:, 2:3] + self.ang_b[None]) * self.ang_f[None]
            d = self.angle_dim
            feats = torch.cat([feats, torch.sin(z[:, :d]), torch.cos(z[:, d:])], dim=1)
        h = torch.relu(torch.nn.functional.linear(feats, self.w1, self.b1))
        h = torch.relu(torch.nn.functional.linear(h, self.w2, self.b2))
        return torch.nn.functional.linear(torch.cat([h, feats], dim=1), self.w3, self.b3)

    # ---- closed-form path (batched baseline): logit and d logit / d pose, no autograd ------------------------------
    def logits_and_input_grad(self, x):
        d = self.angle_dim
        u = (x[:, :2] - self.mean) / self.sigma
        e = torch.addmm(self.be, u, self.wet)
        se, ce = torch.sin(e), torch.cos(e)
        parts = [se[:, :100], ce[:, 100:]] if self.use_cos else [se]
        if d:
            z = (x[:, 2:3] + self.ang_b) * self.ang_f
            sz, cz = torch.sin(z), torch.cos(z)
            parts += [sz[:, :d], cz[:, d:]]
        feats = torch.cat(parts, dim=1)
        a1 = torch.addmm(self.b1, feats, self.w1t)
        h1 = torch.relu(a1)
        a2 = torch.addmm(self.b2, h1, self.w2t)
        h2 = torch.relu(a2)
        w3 = self.w3[0]
        logit = h2 @ w3[:HIDDEN] + feats @ w3[HIDDEN:] + self.b3
        dh2 = w3[:HIDDEN] * (a2 > 0)
        dh1 = (dh2 @ self.w2) * (a1 > 0)
        din = torch.addmm(w3[HIDDEN:], dh1, self.w1)
        n_enc = e.shape[1]
        de = torch.cat([din[:, :100] * ce[:, :100], -din[:, 100:200] * se[:, 100:]], dim=1) if self.use_cos \
            else din[:, :100] * ce
        gxy = (de @ self.we) / self.sigma
        if not d:
            return logit, gxy
        dz = torch.cat([din[:, n_enc:n_enc + d] * cz[:, :d], -din[:, n_enc + d:] * sz[:, d:]], dim=1)
        return logit, torch.cat([gxy, (dz * self.ang_f).sum(1, keepdim=True)], dim=1)


class Scalars(object):
    """Hyper-parameters of the SE(2) planner step (nfop/constrained_nerf_opt_planner.py:13-40 + the Adam group)."""

    def __init__(self, collision_weight=1.0, angle_weight=0.5, constraint_deltas_weight=20.0, multipliers_lr=0.1,
                 collision_multipliers_lr=1e-3, boundary_weight=1.0, collision_beta=1.0, direction_delta_weight=0.0,
                 lr=1e-2, beta1=0.9, beta2=0.9, eps=1e-8, bounds=(0.0, 1.0, 0.0, 1.0), velocity_hessian_weight=0.5,
                 reparam_freq=10):
        self.__dict__.update(locals())
        del self.__dict__["self"]

    @classmethod
    def from_oracle(cls, hp, **kw):
        return cls(hp.collision_weight, hp.angle_weight, hp.constraint_deltas_weight, hp.multipliers_lr,
                   hp.collision_multipliers_lr, hp.boundary_weight, hp.collision_beta, hp.direction_delta_weight,
                   hp.lr, hp.beta1, hp.beta2, hp.eps, hp.bounds, **kw)


def inverse_hessian(n, w):
    """nfop/nerf_opt_planner.py:45-58"""
    k = np.zeros((n, n), np.float32)
    i = np.arange(n)
    k[i, i] = 4
    k[i[1:], i[:-1]] = -2
    k[i[:-1], i[1:]] = -2
    return torch.tensor(np.linalg.inv(w * k + np.eye(n)).astype(np.float32))


def _reparametrize(q, lam, cm):
    """Arc-length reparametrisation of full paths q [B, N+2, 3] with multipliers (constrained:132-171), batched torch."""
    B, n2, _ = q.shape
    N = n2 - 2
    dist = torch.norm(q[:, 1:, :2] - q[:, :-1, :2], dim=2)
    cdf = torch.cat([torch.zeros(B, 1), torch.cumsum(dist / dist.sum(1, keepdim=True), dim=1)], dim=1)
    u = torch.linspace(0, 1, n2)[1:-1].expand(B, N).contiguous()
    idx = torch.searchsorted(cdf, u)
    ia = idx.clamp(max=N + 1)
    ib = (idx - 1).clamp(min=0)
    ca, cb = torch.gather(cdf, 1, ia), torch.gather(cdf, 1, ib)
    tau = (u - cb) / (ca - cb).clamp(min=1e-5)
    qa = torch.gather(q, 1, ia[..., None].expand(B, N, 3))
    qb = torch.gather(q, 1, ib[..., None].expand(B, N, 3))
    out = torch.empty(B, N, 3)
    out[..., :2] = (1 - tau)[..., None] * qb[..., :2] + tau[..., None] * qa[..., :2]
    out[..., 2] = qb[..., 2] + tau * wrap(qa[..., 2] - qb[..., 2])
    cmf = torch.cat([torch.zeros(B, 1), cm, torch.zeros(B, 1)], dim=1)
    new_cm = (1 - tau) * torch.gather(cmf, 1, ib) + tau * torch.gather(cmf, 1, ia)
    lf = torch.cat([lam[:, :1], (lam[:, :-1] + lam[:, 1:]) / 2, lam[:, -1:]], dim=1)
    li = (1 - tau) * torch.gather(lf, 1, ib) + tau * torch.gather(lf, 1, ia)
    new_lam = torch.cat([li[:, :1], (li[:, :-1] + li[:, 1:]) / 2, li[:, -1:]], dim=1)
    return out, new_lam, new_cm


# ----------------------------------------------------------------------------------------------------------------------
class BatchedTorchPlanner(object):
    """(1) strong baseline: B trajectories, closed-form gradients, dense sgemm layers, no autograd."""

    def __init__(self, field, sc, traj, start, goal, lam=None, cm=None, adam_m=None, adam_v=None, adam_step=0,
                 step_count=0):
        f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32)).clone()  # noqa: E731
        self.field, self.sc = field, sc
        self.traj, self.start, self.goal = f32(traj), f32(start), f32(goal)
        B, N, _ = self.traj.shape
        self.lam = f32(lam) if lam is not None else torch.zeros(B, N + 1)
        self.cm = f32(cm) if cm is not None else torch.zeros(B, N)
        self.m = f32(adam_m) if adam_m is not None else torch.zeros(B, N, 3)
        self.v = f32(adam_v) if adam_v is not None else torch.zeros(B, N, 3)
        self.adam_step, self.step_count = int(adam_step), int(step_count)
        self.hinv = inverse_hessian(N, sc.velocity_hessian_weight)
        self.terms = None

    @torch.no_grad()
    def optimize_trajectory(self, t):
        sc, tr = self.sc, self.traj
        B, N, _ = tr.shape
        t = torch.as_tensor(np.asarray(t, np.float32)).reshape(B, N - 1)
        # collision samples + field (constrained:78-85)
        d = tr[:, :-1] - tr[:, 1:]
        d[..., 2] = wrap(d[..., 2])
        pts = tr[:, 1:] + t[..., None] * d
        logit, dl = self.field.logits_and_input_grad(pts.reshape(-1, 3))
        logit, dl = logit.reshape(B, N - 1), dl.reshape(B, N - 1, 3)
        q = torch.cat([self.start[:, None], tr, self.goal[:, None]], dim=1)
        G = torch.zeros_like(q)
        aw = sc.angle_weight
        # distance (constrained:120-130)
        delta = q[:, 1:] - q[:, :-1]
        winding = wrap(delta[..., 2]).sum(1) - q[:, -1, 2] + q[:, 0, 2]
        delta[:, -1, 2] += winding
        delta[..., 2] *= aw
        l_dist = (delta * delta).sum((1, 2))
        gd = 2 * delta
        gd[..., 2] *= aw
        G[:, 1:] += gd
        G[:, :-1] -= gd
        # non-holonomic (constrained:102-109)
        dx, dy, th = q[:, 1:, 0] - q[:, :-1, 0], q[:, 1:, 1] - q[:, :-1, 1], q[..., 2]
        m = th[:, :-1] + wrap(th[:, 1:] - th[:, :-1]) / 2
        sm, cmm = torch.sin(m), torch.cos(m)
        c = dx * sm - dy * cmm
        e = dx * cmm + dy * sm
        g = self.lam + 2 * sc.constraint_deltas_weight * c
        G[:, 1:, 0] += g * sm
        G[:, :-1, 0] -= g * sm
        G[:, 1:, 1] -= g * cmm
        G[:, :-1, 1] += g * cmm
        G[:, :-1, 2] += g * e / 2
        G[:, 1:, 2] += g * e / 2
        # direction / forward-only (constrained:111-118, :93, :98)
        mp = th[:, :-1] + wrap(th[:, :-1] - th[:, 1:]) / 2
        smp, cmp_ = torch.sin(mp), torch.cos(mp)
        r = torch.relu(-(cmp_ * dx + smp * dy))
        hh = 2 * sc.direction_delta_weight * r
        k = smp * dx - cmp_ * dy
        G[:, 1:, 0] -= hh * cmp_
        G[:, :-1, 0] += hh * cmp_
        G[:, 1:, 1] -= hh * smp
        G[:, :-1, 1] += hh * smp
        G[:, :-1, 2] += 1.5 * hh * k
        G[:, 1:, 2] -= 0.5 * hh * k
        # boundary (nerf:171-176)
        lo_x, hi_x, lo_y, hi_y = sc.bounds
        x, y = tr[..., 0], tr[..., 1]
        bx0, bx1, by0, by1 = torch.relu(lo_x - x), torch.relu(x - hi_x), torch.relu(lo_y - y), torch.relu(y - hi_y)
        l_bnd = (bx0 ** 2 + bx1 ** 2 + by0 ** 2 + by1 ** 2).sum(1)
        G[:, 1:-1, 0] += 2 * sc.boundary_weight * (bx1 - bx0)
        G[:, 1:-1, 1] += 2 * sc.boundary_weight * (by1 - by0)
        # collision (constrained:82-89)
        beta = sc.collision_beta
        sp = torch.nn.functional.softplus(logit, beta)
        dsp = torch.where(logit * beta > 20, torch.ones_like(logit), torch.sigmoid(logit * beta))
        tanh_l = torch.tanh(logit)
        cm_i = self.cm[:, 1:] * (1 - t) + self.cm[:, :-1] * t
        gg = (sc.collision_weight * dsp + cm_i * (1 - tanh_l * tanh_l))[..., None] * dl
        G[:, 1:-2] += t[..., None] * gg
        G[:, 2:-1] += (1 - t)[..., None] * gg
        g_cm = torch.zeros_like(self.cm)
        g_cm[:, 1:] += (1 - t) * tanh_l
        g_cm[:, :-1] += t * tanh_l
        l_col, l_cm, l_dir = sp.sum(1), (cm_i * tanh_l).sum(1), (r * r).sum(1)
        total = (l_dist + sc.collision_weight * l_col + (self.lam * c).sum(1) + sc.constraint_deltas_weight * (c * c).sum(1)
                 + sc.boundary_weight * l_bnd + l_cm + sc.direction_delta_weight * l_dir)
        self.terms = dict(total=total, l_dist=l_dist, l_col=l_col, l_cm=l_cm, l_bnd=l_bnd, l_dir=l_dir, c=c)
        # H^-1 g, Adam (torch single-tensor form), multiplier ascent (nerf:151-154, constrained:66-73)
        grad = torch.matmul(self.hinv, G[:, 1:-1])
        self.adam_step += 1
        self.m.lerp_(grad, 1 - sc.beta1)
        self.v.mul_(sc.beta2).addcmul_(grad, grad, value=1 - sc.beta2)
        bc1, bc2 = 1 - sc.beta1 ** self.adam_step, 1 - sc.beta2 ** self.adam_step
        tr.addcdiv_(self.m, (self.v.sqrt() / math.sqrt(bc2)).add_(sc.eps), value=-sc.lr / bc1)
        self.lam += sc.multipliers_lr * c
        self.cm = torch.relu(self.cm + sc.collision_multipliers_lr * g_cm)

    @torch.no_grad()
    def step(self, t):
        self.optimize_trajectory(t)
        if self.step_count % self.sc.reparam_freq == 0:
            q = torch.cat([self.start[:, None], self.traj, self.goal[:, None]], dim=1)
            self.traj, self.lam, self.cm = _reparametrize(q, self.lam, self.cm)
        self.step_count += 1


# ----------------------------------------------------------------------------------------------------------------------
class EagerAutogradPlanner(object):
    """(2) reference-faithful baseline: one trajectory, eager ops + autograd + torch.optim.Adam."""

    def __init__(self, field, sc, traj, start, goal, lam=None, cm=None, adam_m=None, adam_v=None, adam_step=0,
                 step_count=0):
        f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32)).clone()  # noqa: E731
        self.field, self.sc = field, sc
        self.traj = f32(traj).requires_grad_(True)
        n = self.traj.shape[0]
        self.start, self.goal = f32(start).reshape(1, 3), f32(goal).reshape(1, 3)
        self.lam = (f32(lam) if lam is not None else torch.zeros(n + 1)).requires_grad_(True)
        self.cm = (f32(cm) if cm is not None else torch.zeros(n)).requires_grad_(True)
        self.optimizer = torch.optim.Adam([self.traj], lr=sc.lr, betas=(sc.beta1, sc.beta2), eps=sc.eps)
        if adam_step:
            self.optimizer.state[self.traj] = {"step": torch.tensor(float(adam_step)), "exp_avg": f32(adam_m),
                                               "exp_avg_sq": f32(adam_v)}
        self.hinv = inverse_hessian(n, sc.velocity_hessian_weight)
        self.step_count = int(step_count)
        self.last_total = None

    def full(self):
        return torch.cat([self.start, self.traj, self.goal], dim=0)

    def loss(self, t):
        sc, tr = self.sc, self.traj
        t = torch.as_tensor(np.asarray(t, np.float32)).reshape(-1, 1)
        d = tr[:-1] - tr[1:]
        d = torch.cat([d[:, :2], wrap(d[:, 2:3])], dim=1)
        pts = tr[1:] + t * d
        cm_i = self.cm[1:] * (1 - t[:, 0]) + self.cm[:-1] * t[:, 0]
        logit = self.field.logits(pts)
        l_col = torch.sum(torch.nn.functional.softplus(logit, sc.collision_beta))
        l_cm = torch.sum(cm_i * torch.tanh(logit[:, 0]))
        q = self.full()
        dx, dy = q[1:, 0] - q[:-1, 0], q[1:, 1] - q[:-1, 1]
        mean_a = q[:-1, 2] + wrap(q[1:, 2] - q[:-1, 2]) / 2
        c = dx * torch.sin(mean_a) - dy * torch.cos(mean_a)
        mean_b = q[:-1, 2] + wrap(q[:-1, 2] - q[1:, 2]) / 2
        back = torch.relu(-(torch.cos(mean_b) * dx + torch.sin(mean_b) * dy))
        delta = q[1:] - q[:-1]
        winding = (torch.sum(wrap(delta[:, 2])) - q[-1, 2] + q[0, 2]).detach()
        dth = torch.cat([delta[:-1, 2], (delta[-1, 2] + winding)[None]]) * sc.angle_weight
        l_dist = torch.sum(delta[:, :2] ** 2) + torch.sum(dth ** 2)
        lo_x, hi_x, lo_y, hi_y = sc.bounds
        l_bnd = (torch.sum(torch.relu(lo_x - tr[:, 0]) ** 2) + torch.sum(torch.relu(tr[:, 0] - hi_x) ** 2)
                 + torch.sum(torch.relu(lo_y - tr[:, 1]) ** 2) + torch.sum(torch.relu(tr[:, 1] - hi_y) ** 2))
        return (l_dist + sc.collision_weight * l_col + torch.sum(self.lam * c) + sc.constraint_deltas_weight * torch.sum(c ** 2)
                + sc.boundary_weight * l_bnd + l_cm + sc.direction_delta_weight * torch.sum(back ** 2))

    def optimize_trajectory(self, t):
        sc = self.sc
        self.optimizer.zero_grad()
        self.lam.grad = self.cm.grad = None
        total = self.loss(t)
        total.backward()
        self.last_total = total.detach()
        self.traj.grad = self.hinv @ self.traj.grad
        self.optimizer.step()
        with torch.no_grad():
            self.lam += sc.multipliers_lr * self.lam.grad
            self.cm += sc.collision_multipliers_lr * self.cm.grad
            self.cm.copy_(torch.relu(self.cm))

    def step(self, t):
        self.optimize_trajectory(t)
        if self.step_count % self.sc.reparam_freq == 0:
            with torch.no_grad():
                tr, lam, cm = _reparametrize(self.full()[None], self.lam[None], self.cm[None])
                self.traj.data, self.lam.data, self.cm.data = tr[0], lam[0], cm[0]
        self.step_count += 1


# ----------------------------------------------------------------------------------------------------------------------
# timing harness (child process of bench.py; CPU only)
def _straight_lines(starts, goals, n):
    s, g = torch.as_tensor(starts), torch.as_tensor(goals)
    w = torch.linspace(0, 1, n + 2)[1:-1][None, :, None]
    g = torch.cat([g[:, :2], s[:, 2:3] + wrap(g[:, 2:3] - s[:, 2:3])], dim=1)
    return (s[:, None] * (1 - w) + g[:, None] * w).numpy()


def _eager_worker(args):
    flat, cfgv, sc_kw, starts, goals, n, seconds, seed = args
    torch.set_num_threads(1)
    field = Field(flat, cfgv[0], cfgv[1])
    sc = Scalars(**sc_kw)
    rng = np.random.default_rng(seed)
    steps, t_used, b = 0, 0.0, 0
    while t_used < seconds:
        pl = EagerAutogradPlanner(field, sc, _straight_lines(starts[b:b + 1], goals[b:b + 1], n)[0], starts[b], goals[b])
        pl.step(rng.uniform(0, 1, n - 1))    # first call: allocator / autograd warm-up, untimed
        t0 = time.perf_counter()
        for _ in range(20):
            pl.step(rng.uniform(0, 1, n - 1))
        t_used += time.perf_counter() - t0
        steps += 20
        b = (b + 1) % len(starts)
    return steps, t_used


def time_baselines(inp, seconds_batched, seconds_eager, procs):
    z = np.load(inp)
    flat, cfgv, n = z["onf_flat"], z["onf_cfg"], int(z["n_waypoints"])
    starts, goals = z["starts"].astype(np.float32), z["goals"].astype(np.float32)
    sc_kw = {k[3:]: (tuple(float(x) for x in z[k]) if z[k].ndim else float(z[k])) for k in z.files if k.startswith("sc_")}
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = usable_cores()
    out = {"host_cores": visible, "usable_cores": cores, "cpu_model": _cpu_model()}
    # (2) first, on ONE core.  procs > 1 adds a process-parallel figure from a fork pool of single-threaded workers --
    # for CPU-only sessions: on a leased GPU box the process guard counts worker processes against the card's limit,
    # so bench.py asks for the per-core figure only.
    torch.set_num_threads(1)
    procs = max(1, min(procs, cores))
    s1, t1 = _eager_worker((flat, cfgv, sc_kw, starts, goals, n, seconds_eager if procs == 1 else seconds_eager / 2, 0))
    out["reference_faithful"] = {
        "kind": "reference-faithful", "unit": "waypoint-evals/s", "value": s1 * n / t1, "cores": 1,
        "ms_per_step": 1e3 * t1 / s1,
        "sample": "eager autograd + torch.optim.Adam planner (op-for-op restatement of the reference's step), 1 trajectory "
                  "x %d waypoints, frozen field, %d steps on one core, %.1f s" % (n, s1, t1)}
    if procs > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(procs) as pool:
            many = pool.map(_eager_worker, [(flat, cfgv, sc_kw, starts, goals, n, seconds_eager / 2, k) for k in range(procs)])
        out["reference_faithful"]["process_parallel"] = {"procs": procs, "value": sum(s * n / t for s, t in many)}
    # (1) strong baseline on all usable cores
    cores = best_thread_count(cores)
    torch.set_num_threads(cores)
    field = Field(flat, cfgv[0], cfgv[1])
    sc = Scalars(**sc_kw)
    pl = BatchedTorchPlanner(field, sc, _straight_lines(starts, goals, n), starts, goals)
    rng = np.random.default_rng(5)
    B = len(starts)
    draw = lambda: rng.uniform(0, 1, (B, n - 1)).astype(np.float32)  # noqa: E731
    pl.step(draw())
    pl.step(draw())
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds_batched:
        pl.step(draw())
        steps += 1
    dt = time.perf_counter() - t0
    out["strong"] = {"kind": "port", "unit": "waypoint-evals/s", "value": B * n * steps / dt, "cores": cores,
                     "ms_per_step": 1e3 * dt / steps,
                     "sample": "%d trajectories x %d waypoints x %d steps of the same workload, batched autograd-free "
                               "torch-CPU restatement (MKL sgemm), %d threads, %.1f s" % (B, n, steps, cores, dt)}
    return out


def usable_cores():
    """Threads worth starting: the affinity mask, cut down to the cgroup CPU quota when there is one (a leased GPU box
    shows all 256 host threads but grants a share of them; 256 OpenMP threads on a 16-core share run 100x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:            # cgroup v2: "<quota|max> <period>"
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(math.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0:
                n = min(n, max(1, int(math.ceil(quota / period))))
        except (OSError, ValueError):
            pass
    return n


def best_thread_count(limit):
    """No quota visible is not proof of none: time the baseline's dominant sgemm at 8, 16, 32, ... threads (up to
    `limit`) and keep the fastest -- a few hundred milliseconds, and it picks what a CPU user would pick."""
    a, b = torch.randn(65280, 224), torch.randn(224, 100)
    best, best_t = 1, float("inf")
    n = min(8, limit)
    while True:
        torch.set_num_threads(n)
        torch.mm(a, b)
        t0 = time.perf_counter()
        for _ in range(5):
            torch.mm(a, b)
        dt = time.perf_counter() - t0
        if dt < best_t:
            best, best_t = n, dt
        if n >= limit or dt > 3 * best_t:
            break
        n = min(2 * n, limit)
    return best


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--input", required=True)
    ap.add_argument("--seconds-batched", type=float, default=10.0)
    ap.add_argument("--seconds-eager", type=float, default=8.0)
    ap.add_argument("--procs", type=int, default=1)
    a = ap.parse_args()
    json.dump(time_baselines(a.input, a.seconds_batched, a.seconds_eager, a.procs), sys.stdout)
    sys.stdout.write("\n")
