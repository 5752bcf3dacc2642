#!/usr/bin/env python3
"""Generate golden vectors from the reference (MisterMap/pytorch-motion-planner) PyTorch-CPU path.

Runs ONLY in the build container (needs /root/reference, read-only).  The reference is imported
unmodified; two harness-side shims are installed first:
  * stub `pytorch_lightning.utilities[.parsing].AttributeDict` (package absent here)
  * `numpy.bool = bool` (reference uses the alias removed in numpy>=1.24, nerf_opt_planner.py:43)
Random draws are injected by re-seeding torch's CPU generator right before the reference draws
`torch.rand(N-1, 1)` (constrained_nerf_opt_planner.py:78) and replaying the same draw here.

Outputs: small fp32 `.npz` fixtures next to this script.  Nothing from the reference's source text is
written out -- only inputs and the numbers the reference computed from them.

Usage:  MPLBACKEND=Agg python tests/golden/make_golden.py [generator ...]   (no names: all; see GENERATORS)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("NFOPP_REFERENCE", "/root/reference")


class AttributeDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def install_shims():
    for name in ("pytorch_lightning", "pytorch_lightning.utilities", "pytorch_lightning.utilities.parsing"):
        m = types.ModuleType(name)
        m.AttributeDict = AttributeDict
        sys.modules[name] = m
    if not hasattr(np, "bool"):
        np.bool = bool
    sys.path.insert(0, REF)


install_shims()
from neural_field_optimal_planner.collision_checker import (  # noqa: E402
    CircleDirectedCollisionChecker, CircleCollisionChecker, RectangleCollisionChecker)
from neural_field_optimal_planner.onf_model import ONF  # noqa: E402
from neural_field_optimal_planner.planner_factory import PlannerFactory  # noqa: E402
from neural_field_optimal_planner.test_environment_builder import TestEnvironmentBuilder  # noqa: E402
from neural_field_optimal_planner.trajectory_initializer import TrajectoryInitializer  # noqa: E402
from neural_field_optimal_planner.torch_math import wrap_angle  # noqa: E402
from neural_field_optimal_planner.utils.position2 import Position2  # noqa: E402

F32 = np.float32


def params(n=100, **planner_over):
    """Parameter block of scripts/benchmark.py:22-63 (values restated, N overridable)."""
    p = AttributeDict(
        device="cpu", trajectory_length=n,
        collision_model=AttributeDict(mean=0, sigma=1, use_cos=True, bias=True, use_normal_init=True,
                                      angle_encoding=True, name="ONF"),
        trajectory_initializer=AttributeDict(name="TrajectoryInitializer", resolution=0.05),
        collision_optimizer=AttributeDict(lr=5e-2, betas=(0.9, 0.9)),
        trajectory_optimizer=AttributeDict(lr=1e-2, betas=(0.9, 0.9)),
        planner=AttributeDict(name="ConstrainedNERFOptPlanner", trajectory_random_offset=0.02,
                              collision_weight=1, velocity_hessian_weight=0.5, random_field_points=10,
                              init_collision_iteration=0, constraint_deltas_weight=20, multipliers_lr=0.1,
                              init_collision_points=100, reparametrize_trajectory_freq=10,
                              optimize_collision_model_freq=1, angle_weight=0.5, angle_offset=0.3,
                              boundary_weight=1, collision_multipliers_lr=1e-3))
    p.planner.update(planner_over)
    return p


def make_planner(n=100, start=None, goal=None, seed_t=100, seed_np=400, **planner_over):
    torch.random.manual_seed(seed_t)
    np.random.seed(seed_np)
    env = TestEnvironmentBuilder().make_test_environment_with_angles()
    cc = CircleDirectedCollisionChecker(0.3, (0, 3, 0, 3))
    cc.update_obstacle_points(env.obstacle_points)
    cc.update_boundaries(env.bounds)
    planner = PlannerFactory.make_constrained_onf_planner(cc, params(n, **planner_over))
    start = env.start_point if start is None else np.asarray(start, F32)
    goal = env.goal_point if goal is None else np.asarray(goal, F32)
    planner.init(start, goal, env.bounds)
    torch.autograd.set_detect_anomaly(False)
    return planner, env


def flat_params(model):
    return np.concatenate([v.detach().cpu().numpy().reshape(-1) for v in model.state_dict().values()]).astype(F32)


def freeze(planner):
    planner._optimize_collision_model_freq = 10 ** 9
    if planner._step_count % planner._optimize_collision_model_freq == 0:
        planner._step_count = 1


def hyper(planner):
    g = planner._trajectory_optimizer.param_groups[0]
    return dict(
        collision_weight=planner._collision_weight, angle_weight=planner._angle_weight,
        constraint_deltas_weight=planner._constraint_delta_weight, multipliers_lr=planner._multipliers_lr,
        collision_multipliers_lr=planner._collision_multipliers_lr, boundary_weight=planner._boundary_weight,
        collision_beta=planner._collision_beta, direction_delta_weight=planner._direction_delta_weight,
        lr=g["lr"], beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"],
        bounds=np.asarray(planner._random_sample_border, F32))


def npz_hyper(h):
    return {"hp_" + k: np.asarray(v, dtype=np.float64 if np.isscalar(v) else None) for k, v in h.items()}


def state(planner):
    st = planner._trajectory_optimizer.state.get(planner._trajectory, {})
    n, d = planner._trajectory.shape
    return dict(
        traj=planner._trajectory.detach().numpy().copy(),
        start=planner._start_point.detach().numpy().copy()[0],
        goal=planner._goal_point.detach().numpy().copy()[0],
        lam=planner._constraint_multipliers.detach().numpy().copy(),
        cm=planner._collision_multipliers.detach().numpy().copy(),
        adam_m=st["exp_avg"].numpy().copy() if st else np.zeros((n, d), F32),
        adam_v=st["exp_avg_sq"].numpy().copy() if st else np.zeros((n, d), F32),
        adam_step=np.asarray(float(st["step"]) if st else 0.0),
        step_count=np.asarray(planner._step_count))


def draw_t(n, seed):
    torch.random.manual_seed(seed)
    t = torch.rand(n - 1, 1)
    torch.random.manual_seed(seed)  # the reference will now draw the identical tensor
    return t


def collision_positions(planner, t):
    """Harness-side sample points (same inputs the reference forms internally) -- used only to split the
    collision loss into its two summands; the reference's total loss pins their sum."""
    tr = planner._trajectory.detach()
    d = tr[:-1] - tr[1:]
    d[:, 2] = wrap_angle(d[:, 2])
    return tr[1:] + t * d


# ----------------------------------------------------------------------------------------------------------
def g1_onf():
    """ONF logits and input gradients (onf_model.py:33-50) on trained and random-init fields."""
    out = {}
    planner, env = make_planner(100)
    for _ in range(100):
        planner.step()
    model = planner._collision_model
    rng = np.random.default_rng(11)
    x = np.stack([rng.uniform(-0.3, 3.3, 1024), rng.uniform(-0.3, 3.3, 1024), rng.uniform(-4, 4, 1024)], 1).astype(F32)
    xt = torch.tensor(x, requires_grad=True)
    y = model(xt)
    (g,) = torch.autograd.grad(y.sum(), xt)
    out.update(a_params=flat_params(model), a_cfg=np.asarray([0, 1, 1, 1, 1], np.float64),  # mean,sigma,use_cos,bias,angle
               a_x=x, a_logit=y.detach().numpy()[:, 0], a_grad=g.numpy())

    # sigma=10 field on a 100x100 map (scripts/run_bench_mr.py:28-36 style), random init
    torch.random.manual_seed(7)
    m2 = ONF(0, 10, use_cos=True, use_normal_init=True, bias=True, angle_encoding=True)
    x = np.stack([rng.uniform(0, 100, 1024), rng.uniform(0, 100, 1024), rng.uniform(-np.pi, np.pi, 1024)], 1).astype(F32)
    xt = torch.tensor(x, requires_grad=True)
    y = m2(xt)
    (g,) = torch.autograd.grad(y.sum(), xt)
    out.update(b_params=flat_params(m2), b_cfg=np.asarray([0, 10, 1, 1, 1], np.float64),
               b_x=x, b_logit=y.detach().numpy()[:, 0], b_grad=g.numpy())

    # 2-D field of PlannerFactory.make_onf_planner: ONF(1.5, 1) (sin only, no angle) planner_factory.py:53
    torch.random.manual_seed(8)
    m3 = ONF(1.5, 1)
    x = np.stack([rng.uniform(-0.3, 3.3, 512), rng.uniform(-0.3, 3.3, 512)], 1).astype(F32)
    xt = torch.tensor(x, requires_grad=True)
    y = m3(xt)
    (g,) = torch.autograd.grad(y.sum(), xt)
    out.update(c_params=flat_params(m3), c_cfg=np.asarray([1.5, 1, 0, 1, 0], np.float64),
               c_x=x, c_logit=y.detach().numpy()[:, 0], c_grad=g.numpy())
    np.savez_compressed(os.path.join(HERE, "g1_onf.npz"), **out)
    return planner


def terms_and_grads(planner, seed):
    n = planner._trajectory.shape[0]
    t = draw_t(n, seed)
    for p in (planner._trajectory, planner._constraint_multipliers, planner._collision_multipliers):
        p.grad = None
    planner._collision_model.requires_grad_(False)
    loss = planner.trajectory_loss()
    loss.backward()
    with torch.no_grad():
        pos = collision_positions(planner, t)
        logit = planner._collision_model(pos)
        cmi = planner._collision_multipliers[1:] * (1 - t[:, 0]) + planner._collision_multipliers[:-1] * t[:, 0]
        l_col = torch.sum(torch.nn.functional.softplus(logit, planner._collision_beta))
        l_cm = torch.sum(cmi * torch.tanh(logit[:, 0]))
        c = planner.non_holonomic_constraint_deltas()
        d = planner.direction_constraint_deltas()
        l_dist = planner.distance_loss()
        l_bnd = planner.boundary_loss()
    out = dict(t=t.numpy()[:, 0].copy(), total=np.asarray(loss.item()), l_dist=np.asarray(l_dist.item()),
               l_col=np.asarray(l_col.item()), l_cm=np.asarray(l_cm.item()), l_bnd=np.asarray(l_bnd.item()),
               c=c.numpy().copy(), d=d.numpy().copy(), pos=pos.numpy().copy(), logit=logit.numpy()[:, 0].copy(),
               g_traj=planner._trajectory.grad.numpy().copy(),
               g_lam=planner._constraint_multipliers.grad.numpy().copy(),
               g_cm=planner._collision_multipliers.grad.numpy().copy())
    for p in (planner._trajectory, planner._constraint_multipliers, planner._collision_multipliers):
        p.grad = None
    return out


def g2_g3_g6(tag, n, warm, start=None, goal=None, over=None, rollout=(1, 10, 50, 200), push_out=False):
    """Loss terms + grads (constrained:76-130), one optimiser step (nerf:143-155, constrained:63-74),
    K-step frozen-ONF rollouts (nerf:60-71)."""
    planner, env = make_planner(n, start, goal)
    for _ in range(warm):
        planner.step()
    freeze(planner)
    if over:
        for k, v in over.items():
            setattr(planner, k, v)
    if push_out:  # move a few waypoints outside the bounds so boundary_loss is active
        with torch.no_grad():
            planner._trajectory[n // 3, 0] = -0.35
            planner._trajectory[n // 2, 1] = 3.4
    out = {"params": flat_params(planner._collision_model), "cfg": np.asarray([0, 1, 1, 1, 1], np.float64),
           "hinv": planner._inv_hessian.numpy().copy(), "vh_weight": np.asarray(0.5)}
    out.update(npz_hyper(hyper(planner)))
    s0 = state(planner)
    out.update({"s0_" + k: v for k, v in s0.items()})
    out.update({"g2_" + k: v for k, v in terms_and_grads(planner, 5000).items()})

    # G3: one _optimize_trajectory
    t = draw_t(n, 6000)
    planner._optimize_trajectory()
    out["g3_t"] = t.numpy()[:, 0].copy()
    out.update({"g3_" + k: v for k, v in state(planner).items()})

    # G6: rollouts (continue from the post-G3 state), full step() incl. reparametrisation
    ts = []
    done = 0
    for K in rollout:
        while done < K:
            t = draw_t(n, 7000 + done)
            ts.append(t.numpy()[:, 0].copy())
            planner.step()
            done += 1
        out.update({"g6_k%d_" % K + k: v for k, v in state(planner).items()})
    out["g6_t"] = np.stack(ts).astype(F32)
    np.savez_compressed(os.path.join(HERE, "traj_%s.npz" % tag), **out)


def g4_reparam():
    """reparametrize_trajectory in -> out (constrained:132-171) incl. clamp / wrap / endpoint cases."""
    out = {}
    cases = {}
    planner, env = make_planner(100)
    for _ in range(35):
        planner.step()
    freeze(planner)
    rng = np.random.default_rng(5)

    def run(tag):
        with torch.no_grad():
            out["%s_in_traj" % tag] = planner._trajectory.detach().numpy().copy()
            out["%s_in_lam" % tag] = planner._constraint_multipliers.detach().numpy().copy()
            out["%s_in_cm" % tag] = planner._collision_multipliers.detach().numpy().copy()
            out["%s_start" % tag] = planner._start_point.numpy()[0].copy()
            out["%s_goal" % tag] = planner._goal_point.numpy()[0].copy()
            planner.reparametrize_trajectory()
            out["%s_out_traj" % tag] = planner._trajectory.detach().numpy().copy()
            out["%s_out_lam" % tag] = planner._constraint_multipliers.detach().numpy().copy()
            out["%s_out_cm" % tag] = planner._collision_multipliers.detach().numpy().copy()

    with torch.no_grad():
        planner._collision_multipliers.data = torch.tensor(rng.uniform(0, 0.2, 100).astype(F32))
        planner._constraint_multipliers.data = torch.tensor(rng.normal(0, 0.3, 101).astype(F32))
    run("mid")
    with torch.no_grad():  # uneven spacing + angles that straddle the +-pi cut
        tr = planner._trajectory
        s = torch.tensor(np.sort(rng.uniform(0, 1, 100)).astype(F32))
        tr[:, 0] = 0.5 + 2.0 * s ** 2
        tr[:, 1] = 0.5 + 1.0 * s
        tr[:, 2] = wrap_angle(torch.tensor(np.linspace(2.6, 3.9, 100).astype(F32)))
        planner._collision_multipliers.data = torch.tensor(rng.uniform(0, 0.2, 100).astype(F32))
        planner._constraint_multipliers.data = torch.tensor(rng.normal(0, 0.3, 101).astype(F32))
    run("wrap")
    with torch.no_grad():  # repeated waypoints => zero-length segments => denominator clamp
        tr = planner._trajectory
        tr[40:60] = tr[40].clone()
        tr[95:] = planner._goal_point
    run("clamp")
    np.savez_compressed(os.path.join(HERE, "g4_reparam.npz"), **out)

    # update_goal_point / update_start_point (constrained:178-194)
    out = {}
    planner, env = make_planner(100)
    for _ in range(25):
        planner.step()
    freeze(planner)
    out["in_traj"] = planner._trajectory.detach().numpy().copy()
    out["in_lam"] = planner._constraint_multipliers.detach().numpy().copy()
    out["in_cm"] = planner._collision_multipliers.detach().numpy().copy()
    out["start"] = planner._start_point.numpy()[0].copy()
    out["goal"] = planner._goal_point.numpy()[0].copy()
    new_goal = np.asarray([2.3, 1.7, 0.4], F32)
    planner.update_goal_point(new_goal)
    out["new_goal"] = new_goal
    out["goal_out_traj"] = planner._trajectory.detach().numpy().copy()
    out["goal_out_lam"] = planner._constraint_multipliers.detach().numpy().copy()
    out["goal_out_cm"] = planner._collision_multipliers.detach().numpy().copy()
    new_start = np.asarray([0.7, 0.55, -0.2], F32)
    planner.update_start_point(new_start)
    out["new_start"] = new_start
    out["start_out_traj"] = planner._trajectory.detach().numpy().copy()
    out["start_out_lam"] = planner._constraint_multipliers.detach().numpy().copy()
    out["start_out_cm"] = planner._collision_multipliers.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "g4_update_endpoints.npz"), **out)


def g5_hinv():
    """float64-inverse-rounded-to-fp32 preconditioner (nerf_opt_planner.py:45-58)."""
    planner, _ = make_planner(100)
    out = {}
    for n in (16, 100, 256, 512):
        for w in (0.5, 3.0):
            h = planner._calculate_inv_hessian(n, w).numpy()
            key = "n%d_w%s" % (n, str(w).replace(".", "p"))
            if n <= 100:
                out[key] = h
            else:  # keep fixtures small: a band of half-width 64 plus the energy outside it
                band = np.zeros((n, 129), F32)
                for i in range(n):
                    lo, hi = max(0, i - 64), min(n, i + 65)
                    band[i, lo - i + 64:hi - i + 64] = h[i, lo:hi]
                outside = np.abs(h).sum() - np.abs(band).sum()
                out[key + "_band64"] = band
                out[key + "_abs_outside"] = np.asarray(outside)
    np.savez_compressed(os.path.join(HERE, "g5_hinv.npz"), **out)


def g7_onf_train():
    """One _optimize_collision_model step with injected samples (nerf_opt_planner.py:76-91)."""
    planner, env = make_planner(100)
    for _ in range(30):
        planner.step()
    model = planner._collision_model
    opt = planner._collision_optimizer
    rng = np.random.default_rng(21)
    x = np.stack([rng.uniform(-0.1, 3.1, 209), rng.uniform(-0.1, 3.1, 209), rng.uniform(-3.3, 3.3, 209)], 1)
    out = {"cfg": np.asarray([0, 1, 1, 1, 1], np.float64), "x": x.astype(np.float64)}
    names = list(model.state_dict().keys())
    plist = dict(model.named_parameters())
    out["params_before"] = flat_params(model)
    out["adam_m_before"] = np.concatenate([opt.state[plist[k]]["exp_avg"].numpy().reshape(-1) for k in names])
    out["adam_v_before"] = np.concatenate([opt.state[plist[k]]["exp_avg_sq"].numpy().reshape(-1) for k in names])
    out["adam_step_before"] = np.asarray(float(opt.state[plist[names[0]]]["step"]))
    g = opt.param_groups[0]
    out["lr"], out["beta1"], out["beta2"], out["eps"] = (np.asarray(g["lr"]), np.asarray(g["betas"][0]),
                                                         np.asarray(g["betas"][1]), np.asarray(g["eps"]))
    # loss/grad without stepping (same computation the reference performs inside the step)
    model.requires_grad_(True)
    opt.zero_grad()
    pred = planner._calculate_predicted_collision(x)
    truth = planner._calculate_truth_collision(x)
    tt = torch.tensor(truth.astype(np.float32)[:, None])
    loss = planner._collision_loss_function(pred, tt)
    loss.backward()
    out["labels"] = truth.astype(F32)
    out["loss"] = np.asarray(loss.item())
    out["logit"] = pred.detach().numpy()[:, 0].copy()
    out["grad"] = np.concatenate([plist[k].grad.numpy().reshape(-1) for k in names])
    opt.zero_grad()
    planner._optimize_collision_model(x)
    out["params_after"] = flat_params(model)
    out["adam_m_after"] = np.concatenate([opt.state[plist[k]]["exp_avg"].numpy().reshape(-1) for k in names])
    out["adam_v_after"] = np.concatenate([opt.state[plist[k]]["exp_avg_sq"].numpy().reshape(-1) for k in names])
    np.savez_compressed(os.path.join(HERE, "g7_onf_train.npz"), **out)


def g8_batch():
    """B independent reference problems sharing one frozen ONF (batch axis is new: SURVEY fact 2)."""
    base, env = make_planner(100)
    for _ in range(60):
        base.step()
    sd = {k: v.clone() for k, v in base._collision_model.state_dict().items()}
    starts = np.asarray([[0.5, 0.5, 0.0], [0.4, 2.6, -1.0], [2.6, 0.4, 2.5], [0.3, 1.4, 3.0]], F32)
    goals = np.asarray([[2.5, 1.5, 0.0], [2.7, 0.5, 0.5], [0.4, 2.7, -2.8], [2.8, 2.4, -3.0]], F32)
    K = 12
    n = 100
    out = {"params": flat_params(base._collision_model), "cfg": np.asarray([0, 1, 1, 1, 1], np.float64),
           "starts": starts, "goals": goals, "hinv": base._inv_hessian.numpy().copy()}
    trajs0, trajs, lams, cms, ts = [], [], [], [], []
    for b in range(4):
        p, _ = make_planner(n, starts[b], goals[b])
        p._collision_model.load_state_dict(sd)
        freeze(p)
        if b == 0:
            out.update(npz_hyper(hyper(p)))
        trajs0.append(p._trajectory.detach().numpy().copy())
        tb = []
        for k in range(K):
            t = draw_t(n, 9000 + 100 * b + k)
            tb.append(t.numpy()[:, 0].copy())
            p.step()
        ts.append(np.stack(tb))
        trajs.append(p._trajectory.detach().numpy().copy())
        lams.append(p._constraint_multipliers.detach().numpy().copy())
        cms.append(p._collision_multipliers.detach().numpy().copy())
    out.update(traj0=np.stack(trajs0), t=np.stack(ts).astype(F32), traj=np.stack(trajs), lam=np.stack(lams),
               cm=np.stack(cms), steps=np.asarray(K))
    np.savez_compressed(os.path.join(HERE, "g8_batch.npz"), **out)


def g9_full_steps():
    """Whole `.step()` sequence with ONF learning on and the reference's own RNG call order
    (scripts/benchmark.py configuration, seeds torch 100 / numpy 400)."""
    planner, env = make_planner(100)
    out = {"obstacles": env.obstacle_points.astype(np.float64), "bounds": np.asarray(env.bounds, np.float64),
           "start": env.start_point, "goal": env.goal_point,
           "params0": flat_params(planner._collision_model), "traj0": planner._trajectory.detach().numpy().copy()}
    K = 6
    for k in range(K):
        planner.step()
        out["k%d_traj" % k] = planner._trajectory.detach().numpy().copy()
        out["k%d_checked" % k] = planner.checked_positions.as_vec().astype(np.float64)
        out["k%d_truth" % k] = np.asarray(planner.truth_collision).astype(np.uint8)
        out["k%d_params" % k] = flat_params(planner._collision_model)
        out["k%d_lam" % k] = planner._constraint_multipliers.detach().numpy().copy()
        out["k%d_cm" % k] = planner._collision_multipliers.detach().numpy().copy()
    out["steps"] = np.asarray(K)
    np.savez_compressed(os.path.join(HERE, "g9_full_steps.npz"), **out)


def g10_planner2d():
    """NERFOptPlanner (2-D, planner_factory.py:50-59): loss/grad, one step, reparametrisation."""
    torch.random.manual_seed(100)
    np.random.seed(400)
    env = TestEnvironmentBuilder().make_test_environment()
    cc = CircleCollisionChecker(0.3, (0, 3, 0, 3))
    cc.update_obstacle_points(env.obstacle_points)
    planner = PlannerFactory.make_onf_planner(cc)
    planner._init_collision_iteration = 40  # keep the fixture quick; semantics unchanged
    planner.init(env.start_point, env.goal_point, env.bounds)
    torch.autograd.set_detect_anomaly(False)
    for _ in range(15):
        planner.step()
    freeze(planner)
    n = 100
    out = {"params": flat_params(planner._collision_model), "cfg": np.asarray([1.5, 1, 0, 1, 0], np.float64),
           "hinv": planner._inv_hessian.numpy().copy(), "vh_weight": np.asarray(3.0),
           "collision_weight": np.asarray(planner._collision_weight),
           "start": planner._start_point.numpy()[0].copy(), "goal": planner._goal_point.numpy()[0].copy(),
           "bounds": np.asarray(env.bounds, F32)}
    g = planner._trajectory_optimizer.param_groups[0]
    out.update(lr=np.asarray(g["lr"]), beta1=np.asarray(g["betas"][0]), beta2=np.asarray(g["betas"][1]),
               eps=np.asarray(g["eps"]))
    st = planner._trajectory_optimizer.state[planner._trajectory]
    out.update(s0_traj=planner._trajectory.detach().numpy().copy(), s0_m=st["exp_avg"].numpy().copy(),
               s0_v=st["exp_avg_sq"].numpy().copy(), s0_step=np.asarray(float(st["step"])))
    # t comes from numpy RNG here (nerf_opt_planner.py:113-117)
    np.random.seed(123)
    t = np.random.rand(n - 1).astype(F32)
    np.random.seed(123)
    planner._trajectory.grad = None
    loss = planner.trajectory_loss()
    loss.backward()
    out.update(g2_t=t, g2_total=np.asarray(loss.item()), g2_grad=planner._trajectory.grad.numpy().copy())
    planner._trajectory.grad = None
    np.random.seed(124)
    t = np.random.rand(n - 1).astype(F32)
    np.random.seed(124)
    planner._optimize_trajectory()
    out.update(g3_t=t, g3_traj=planner._trajectory.detach().numpy().copy(), g3_m=st["exp_avg"].numpy().copy(),
               g3_v=st["exp_avg_sq"].numpy().copy())
    with torch.no_grad():
        planner.reparametrize_trajectory()
    out["g4_traj"] = planner._trajectory.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "g10_planner2d.npz"), **out)


def g11_init_and_checkers():
    """TrajectoryInitializer (trajectory_initializer.py:12-43) and ground-truth checkers
    (collision_checker/*.py) known answers."""
    out = {}
    ti = TrajectoryInitializer(None)
    cases = np.asarray([[0.5, 0.5, 0.0, 2.5, 1.5, 0.0], [0.4, 2.6, 3.0, 2.7, 0.5, -3.0],
                        [1.0, 1.0, -2.0, 1.0, 2.0, 2.5]], F32)
    res = []
    for c in cases:
        tr = torch.zeros(50, 3)
        ti.initialize_trajectory(tr, torch.tensor(c[None, :3]), torch.tensor(c[None, 3:]))
        res.append(tr.numpy().copy())
    out["init_cases"], out["init_traj"] = cases, np.stack(res)
    env = TestEnvironmentBuilder().make_car_environment()
    rng = np.random.default_rng(3)
    x = np.stack([rng.uniform(-0.2, 3.2, 400), rng.uniform(-0.2, 3.2, 400), rng.uniform(-4, 4, 400)], 1)
    rc = RectangleCollisionChecker((-0.3, 0.2, -0.3, 0.2), (0, 3, 0, 3))
    rc.update_obstacle_points(env.obstacle_points)
    out["car_obstacles"] = env.obstacle_points
    out["poses"] = x
    out["rect_truth"] = rc.check_collision(Position2.from_vec(x)).astype(np.uint8)
    env2 = TestEnvironmentBuilder().make_test_environment_with_angles()
    cd = CircleDirectedCollisionChecker(0.3, (0, 3, 0, 3))
    cd.update_obstacle_points(env2.obstacle_points)
    out["corridor_obstacles"] = env2.obstacle_points
    out["circle_truth"] = cd.check_collision(Position2.from_vec(x)).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "g11_init_checkers.npz"), **out)


def g12_init_direction_and_postprocess():
    """TrajectoryInitializer with init_angles_with_trajectory=True
    (trajectory_initializer.py:31-45; its debug prints are swallowed) and ros/path_postprocessor.py:13-69 on
    fp32 planner paths (the ROS adapter hands `Position2.from_vec(planner.get_path())`, goal_planner_adapter.py:56-60)."""
    import contextlib
    import io
    from neural_field_optimal_planner.ros.path_postprocessor import PathPostprocessor
    out = {}
    ti = TrajectoryInitializer(None, init_angles_with_trajectory=True)
    cases = np.asarray([[0.5, 0.5, 0.0, 2.5, 1.5, 0.0], [0.4, 2.6, 3.0, 2.7, 0.5, -3.0],
                        [1.0, 1.0, -2.0, 1.0, 2.0, 2.5], [2.0, 2.0, 1.0, -1.0, 0.5, -1.0]], F32)
    for n in (50, 51):
        res = []
        for c in cases:
            tr = torch.zeros(n, 3)
            with contextlib.redirect_stdout(io.StringIO()):
                ti.initialize_trajectory(tr, torch.tensor(c[None, :3]), torch.tensor(c[None, 3:]))
            res.append(tr.numpy().copy())
        out["dir_traj_n%d" % n] = np.stack(res)
    out["dir_cases"] = cases

    # post-processor inputs: (a) a planned path (final path of the G9 run), (b) a smooth S-curve with a reversing
    # first few poses (direction flip trimmed), (c) a path with repeated poses (filtered), (d) a short 3-pose path
    rng = np.random.default_rng(12)
    paths = []
    g9 = np.load(os.path.join(HERE, "g9_full_steps.npz"))
    paths.append(np.concatenate([g9["start"].reshape(1, 3), g9["k5_traj"], g9["goal"].reshape(1, 3)]).astype(F32))
    s = np.linspace(0, 1, 120)
    pb = np.stack([3 * s, np.sin(3 * s), np.arctan2(3 * np.cos(3 * s), 3.0)], 1)
    pb[:3, 0] = pb[3, 0] + np.asarray([0.12, 0.08, 0.04])            # first poses lie AHEAD: backwards start
    paths.append(pb.astype(F32))
    pc = np.stack([2 * s, 0.5 * s ** 2, 0.5 * s + 3.0], 1)            # heading near +-pi: unfold matters
    pc = np.repeat(pc, 2, axis=0)[:200]
    pc[1::2, :2] += rng.normal(0, 2e-4, (100, 2))
    paths.append(pc.astype(F32))
    paths.append(np.asarray([[0, 0, 0], [0.5, 0.1, 0.3], [1.0, 0.4, 0.6]], F32))
    pe = np.stack([20 * s, 10 * np.sin(2 * s), rng.uniform(-3, 3, 120)], 1)   # long path, wild headings
    paths.append(pe.astype(F32))
    pp = PathPostprocessor()
    for i, path in enumerate(paths):
        res = pp.process(Position2.from_vec(path.copy())).as_vec()
        out["post_in_%d" % i] = path
        out["post_out_%d" % i] = np.asarray(res)
    pp2 = PathPostprocessor(minimal_distance=0.01, distance_step=0.11)
    out["post_out_alt_1"] = np.asarray(pp2.process(Position2.from_vec(paths[1].copy())).as_vec())
    out["post_alt_params"] = np.asarray([0.01, 0.11])
    np.savez_compressed(os.path.join(HERE, "g12_init_dir_postprocess.npz"), **out)


# ----------------------------------------------------------------------------------------------------------
# The settings the benchmark runs: scripts/run_bench_mr.py:19-63 hyper block on the random-disc map of bench.py
BENCH_BOUNDS = (0.0, 100.0, 0.0, 100.0)


def bench_discs():
    """bench.py make_environment(): 300 discs r=1.5 on 100 m x 100 m."""
    return np.random.default_rng(1234).uniform(5, 95, (300, 2)), 1.5


def benchmr_params(n, init_iters, init_points):
    """Parameter block of scripts/run_bench_mr.py:19-63 (values restated).  The A* seed is out of scope (bench-mr is
    absent), so the stock TrajectoryInitializer is used; the field is pre-fitted by the reference's own
    `_init_collision_model` (nerf_opt_planner.py:197-200) on uniform map samples."""
    return AttributeDict(
        device="cpu", trajectory_length=n,
        trajectory_initializer=AttributeDict(name="TrajectoryInitializer", resolution=0.5,
                                             init_angles_with_trajectory=False),
        collision_model=AttributeDict(mean=0, sigma=10, use_cos=True, bias=True, use_normal_init=True,
                                      angle_encoding=True, name="ONF"),
        collision_optimizer=AttributeDict(lr=2e-2, betas=(0.9, 0.9)),
        trajectory_optimizer=AttributeDict(lr=5e-2, betas=(0.9, 0.9)),
        planner=AttributeDict(name="ConstrainedNERFOptPlanner", trajectory_random_offset=0.02, collision_weight=100,
                              velocity_hessian_weight=0.5, random_field_points=10, init_collision_iteration=init_iters,
                              constraint_deltas_weight=100, multipliers_lr=0.1, init_collision_points=init_points,
                              reparametrize_trajectory_freq=10, optimize_collision_model_freq=1, angle_weight=5,
                              angle_offset=0.3, boundary_weight=1, direction_delta_weight=100,
                              collision_multipliers_lr=1e-3, collision_beta=10))


def make_benchmr_planner(n, start, goal, init_iters=300, init_points=2000):
    torch.random.manual_seed(100)
    np.random.seed(400)
    discs, radius = bench_discs()
    cc = CircleDirectedCollisionChecker(radius, BENCH_BOUNDS)
    cc.update_obstacle_points(discs)
    planner = PlannerFactory.make_constrained_onf_planner(cc, benchmr_params(n, init_iters, init_points))
    planner.init(np.asarray(start, F32), np.asarray(goal, F32), BENCH_BOUNDS)
    torch.autograd.set_detect_anomaly(False)
    return planner, cc


def g13_benchmr(tag, n, warm, start, goal, rollout, init_points, learn_during_warm):
    """G2/G3/G6 on the benchmarked settings: sigma=10 field AFTER fitting, w_col 100, beta 10, w_dir 100, aw 5,
    lr 5e-2, random-disc map, N = 256 / 512.  The field is fitted by the reference's `_init_collision_model` (300
    fits on `init_points` uniform map poses); then `warm` planner steps either with the field FROZEN (bench.py's
    headline workload: pre-fitted frozen field) or with ONF learning on (the continuous-learning workload)."""
    planner, cc = make_benchmr_planner(n, start, goal, init_points=init_points)
    if not learn_during_warm:
        freeze(planner)
    for _ in range(warm):
        if not learn_during_warm:
            draw_t(n, 4100 + _)
        planner.step()
    freeze(planner)
    discs, radius = bench_discs()
    out = {"params": flat_params(planner._collision_model), "cfg": np.asarray([0, 10, 1, 1, 1], np.float64),
           "vh_weight": np.asarray(0.5), "discs": discs, "radius": np.asarray(radius),
           "fit_bce": np.asarray(field_bce(planner, cc))}
    out.update(npz_hyper(hyper(planner)))
    out.update({"s0_" + k: v for k, v in state(planner).items()})
    out.update({"g2_" + k: v for k, v in terms_and_grads(planner, 5100).items()})
    t = draw_t(n, 6100)
    planner._optimize_trajectory()
    out["g3_t"] = t.numpy()[:, 0].copy()
    out.update({"g3_" + k: v for k, v in state(planner).items()})
    ts, done = [], 0
    for K in rollout:
        while done < K:
            t = draw_t(n, 7100 + done)
            ts.append(t.numpy()[:, 0].copy())
            planner.step()
            done += 1
        out.update({"g6_k%d_" % K + k: v for k, v in state(planner).items()})
    out["g6_t"] = np.stack(ts).astype(F32)
    np.savez_compressed(os.path.join(HERE, "traj_benchmr_%s.npz" % tag), **out)
    return planner


def field_bce(planner, cc):
    """How well the fitted field separates the map: BCE on 4096 uniform poses (diagnostic stored with the fixture)."""
    rng = np.random.default_rng(99)
    x = np.concatenate([rng.uniform(0, 100, (4096, 2)), rng.uniform(0, 2 * np.pi, (4096, 1))], 1)
    with torch.no_grad():
        logit = planner._collision_model(torch.tensor(x.astype(F32)))
        y = torch.tensor(cc.check_collision(Position2.from_vec(x)).astype(F32)[:, None])
        return float(torch.nn.functional.binary_cross_entropy_with_logits(logit, y))


def g14_benchmr_batch(base):
    """Small batch on the benchmarked settings: B = 4 independent reference problems (N = 256) sharing the fitted
    sigma=10 field of `base`, 12 frozen-field steps each with injected t (two reparametrisations)."""
    sd = {k: v.clone() for k, v in base._collision_model.state_dict().items()}
    starts = np.asarray([[8, 12, 0.3], [92, 9, 2.4], [50, 96, -1.6], [6, 55, 3.0]], F32)
    goals = np.asarray([[90, 86, -2.5], [10, 90, 0.7], [47, 4, -1.5], [95, 48, -3.0]], F32)
    K, n = 12, 256
    out = {"params": flat_params(base._collision_model), "cfg": np.asarray([0, 10, 1, 1, 1], np.float64),
           "starts": starts, "goals": goals}
    trajs0, ts = [], []
    snaps = {k: dict(traj=[], lam=[], cm=[]) for k in (1, 3, K)}   # early steps are gated tightly, the last loosely
    for b in range(4):
        p, _ = make_benchmr_planner(n, starts[b], goals[b], init_iters=0)
        p._collision_model.load_state_dict(sd)
        freeze(p)
        if b == 0:
            out.update(npz_hyper(hyper(p)))
        trajs0.append(p._trajectory.detach().numpy().copy())
        tb = []
        for k in range(K):
            t = draw_t(n, 9500 + 100 * b + k)
            tb.append(t.numpy()[:, 0].copy())
            p.step()
            if k + 1 in snaps:
                snaps[k + 1]["traj"].append(p._trajectory.detach().numpy().copy())
                snaps[k + 1]["lam"].append(p._constraint_multipliers.detach().numpy().copy())
                snaps[k + 1]["cm"].append(p._collision_multipliers.detach().numpy().copy())
        ts.append(np.stack(tb))
    out.update(traj0=np.stack(trajs0), t=np.stack(ts).astype(F32), steps=np.asarray(K), snapshots=np.asarray(sorted(snaps)))
    for k, d in snaps.items():
        out.update({"k%d_%s" % (k, name): np.stack(v) for name, v in d.items()})
    np.savez_compressed(os.path.join(HERE, "g14_benchmr_batch.npz"), **out)


def g15_full_steps_n256():
    """BASELINE configs[1]: the drop-in `.step()` with ONF learning on, 1 trajectory x 256 waypoints, corridor
    environment (scripts/benchmark.py configuration at N = 256, seeds torch 100 / numpy 400).  Field weights are
    stored for the first and last step only (132 KB each)."""
    planner, env = make_planner(256)
    out = {"obstacles": env.obstacle_points.astype(np.float64), "bounds": np.asarray(env.bounds, np.float64),
           "start": env.start_point, "goal": env.goal_point,
           "params0": flat_params(planner._collision_model), "traj0": planner._trajectory.detach().numpy().copy()}
    K = 6
    for k in range(K):
        planner.step()
        out["k%d_traj" % k] = planner._trajectory.detach().numpy().copy()
        out["k%d_checked" % k] = planner.checked_positions.as_vec().astype(np.float64)
        out["k%d_truth" % k] = np.asarray(planner.truth_collision).astype(np.uint8)
        if k in (0, K - 1):
            out["k%d_params" % k] = flat_params(planner._collision_model)
        out["k%d_lam" % k] = planner._constraint_multipliers.detach().numpy().copy()
        out["k%d_cm" % k] = planner._collision_multipliers.detach().numpy().copy()
    out["steps"] = np.asarray(K)
    np.savez_compressed(os.path.join(HERE, "g15_full_steps_n256.npz"), **out)


def g18_run_planner_script():
    """BASELINE configs[0] as scripts/run_planner.py:10-66 runs it: make_car_environment(), the off-centre
    RectangleCollisionChecker((-0.3, 0.2, -0.3, 0.2), (0, 3, 0, 3)) with only update_obstacle_points called on it, seeds
    torch 100 / numpy 400, the script's parameter block (= params(100)), six full `.step()`s with ONF learning."""
    torch.random.manual_seed(100)
    np.random.seed(400)
    env = TestEnvironmentBuilder().make_car_environment()
    cc = RectangleCollisionChecker((-0.3, 0.2, -0.3, 0.2), (0, 3, 0, 3))
    cc.update_obstacle_points(env.obstacle_points)
    planner = PlannerFactory.make_constrained_onf_planner(cc, params(100))
    planner.init(env.start_point, env.goal_point, env.bounds)
    out = {"obstacles": env.obstacle_points.astype(np.float64), "bounds": np.asarray(env.bounds, np.float64),
           "start": env.start_point, "goal": env.goal_point, "box": np.asarray((-0.3, 0.2, -0.3, 0.2), np.float64),
           "checker_bounds": np.asarray((0, 3, 0, 3), np.float64),
           "params0": flat_params(planner._collision_model), "traj0": planner._trajectory.detach().numpy().copy()}
    K = 6
    for k in range(K):
        planner.step()
        out["k%d_traj" % k] = planner._trajectory.detach().numpy().copy()
        out["k%d_checked" % k] = planner.checked_positions.as_vec().astype(np.float64)
        out["k%d_truth" % k] = np.asarray(planner.truth_collision).astype(np.uint8)
        if k in (0, K - 1):
            out["k%d_params" % k] = flat_params(planner._collision_model)
        out["k%d_lam" % k] = planner._constraint_multipliers.detach().numpy().copy()
        out["k%d_cm" % k] = planner._collision_multipliers.detach().numpy().copy()
    out["steps"] = np.asarray(K)
    np.savez_compressed(os.path.join(HERE, "g18_run_planner_script.npz"), **out)


def corridor_grid(rows=100, cols=100, radius=3, seed=3, walkers=5, steps=150):
    """Stand-in for bench-mr's corridor grid generator (absent): random-walk corridors of the given radius carved out
    of a fully occupied 100 x 100 grid.  Our own generator -- the GRID is the committed fixture."""
    rng = np.random.default_rng(seed)
    grid = np.full((rows, cols), 255, np.uint8)
    yy, xx = np.mgrid[0:rows, 0:cols]
    for _ in range(walkers):
        p = rng.uniform(10, 90, 2)
        heading = rng.uniform(0, 2 * np.pi)
        for _ in range(steps):
            grid[(xx - p[0]) ** 2 + (yy - p[1]) ** 2 <= radius * radius] = 0
            heading += rng.normal(0, 0.35)
            p = p + 1.0 * np.asarray([np.cos(heading), np.sin(heading)])
            if not (6 <= p[0] <= cols - 7 and 6 <= p[1] <= rows - 7):
                heading += np.pi / 2 + rng.uniform(0, np.pi)
                p = np.clip(p, [6, 6], [cols - 7, rows - 7])
    return grid


def g16_grid_checker():
    """Occupancy-grid ground truth for BASELINE configs[3]: the committed 100 x 100 grid and the labels the
    reference's own `MapCollisionChecker` (notebooks/onf_planner_image_map.ipynb, cell 2) gives for 6000 poses.
    The cell's source is read from the reference at generation time and exec'd as is (it is numpy-only apart from
    `draw_poly`, which needs cv2 and is never called); nothing of it is written out."""
    import json
    from dataclasses import dataclass
    from neural_field_optimal_planner.collision_checker import CollisionChecker
    with open(os.path.join(REF, "notebooks", "onf_planner_image_map.ipynb")) as f:
        cell = "".join(json.load(f)["cells"][2]["source"])
    # the cell is executed: pin its content (the reference tree is untrusted input; the fixture is only regenerated from the
    # very text these labels were made from)
    import hashlib
    digest = hashlib.sha256(cell.encode()).hexdigest()
    if digest != "d54fd758d908a4202f4d8899fcd433fb51cd10cb44177ca37c25b5fce7c70fc5":
        raise RuntimeError("notebooks/onf_planner_image_map.ipynb cell 2 changed (sha256 %s): review it before executing" % digest)
    ns = {"np": np, "dataclass": dataclass, "CollisionChecker": CollisionChecker}
    exec(compile(cell, "onf_planner_image_map.ipynb#cell2", "exec"), ns)
    grid = corridor_grid()
    out = {"grid": grid}
    rng = np.random.default_rng(16)
    for tag, (ox, oy, cell_size) in {"unit": (0.0, 0.0, 1.0), "fine": (-20.0, -10.0, 0.4)}.items():
        rows, cols = grid.shape
        mi = ns["MapImage"](cols=cols, rows=rows, origin_x=ox, origin_y=oy, cell_size=cell_size,
                            map_image=grid.astype(np.float64))
        cc = ns["MapCollisionChecker"](mi, (ox, ox + cols * cell_size, oy, oy + rows * cell_size))
        x = np.concatenate([rng.uniform(ox - 3 * cell_size, ox + (cols + 3) * cell_size, (3000, 1)),
                            rng.uniform(oy - 3 * cell_size, oy + (rows + 3) * cell_size, (3000, 1)),
                            rng.uniform(-np.pi, np.pi, (3000, 1))], 1)
        # poses on and right next to cell edges (the truncating cast decides these) and around the outer rim
        k = rng.integers(-1, cols + 1, (3000, 2)).astype(np.float64)
        edge = np.stack([ox + (k[:, 0] + 0.5) * cell_size + rng.choice([-1e-3, 0, 1e-3], 3000) * cell_size,
                         oy + (k[:, 1] + 0.5) * cell_size + rng.choice([-1e-3, 0, 1e-3], 3000) * cell_size,
                         np.zeros(3000)], 1)
        x = np.concatenate([x, edge]).astype(F32).astype(np.float64)   # exactly representable in fp32
        out[tag + "_geom"] = np.asarray([ox, oy, cell_size])
        out[tag + "_poses"] = x.astype(F32)
        out[tag + "_truth"] = cc.check_collision(Position2.from_vec(x)).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "g16_grid_checker.npz"), **out)


GENERATORS = {}


def _register():
    GENERATORS.update(
        g1=g1_onf,
        traj_n100_default=lambda: g2_g3_g6("n100_default", 100, 60),
        traj_n100_hard=lambda: g2_g3_g6("n100_hard", 100, 60, start=[0.5, 0.5, 2.9], goal=[2.5, 1.5, -2.9],
                                        rollout=(1, 10, 50),
                                        over=dict(_collision_weight=3.0, _direction_delta_weight=7.0, _collision_beta=2.0),
                                        push_out=True),
        traj_n256_default=lambda: g2_g3_g6("n256_default", 256, 40, rollout=(1, 10)),
        g4=g4_reparam, g5=g5_hinv, g7=g7_onf_train, g8=g8_batch, g9=g9_full_steps, g10=g10_planner2d,
        g11=g11_init_and_checkers, g12=g12_init_direction_and_postprocess,
        benchmr=lambda: g14_benchmr_batch(g13_benchmr("n256", 256, 60, [8, 12, 0.3], [90, 86, -2.5], (1, 10, 50), 4096, False)),
        benchmr_n512=lambda: g13_benchmr("n512", 512, 40, [92, 9, 2.4], [10, 90, 0.7], (1, 10), 2000, True),
        g15=g15_full_steps_n256, g16=g16_grid_checker, g18=g18_run_planner_script)


if __name__ == "__main__":
    torch.set_num_threads(1)
    _register()
    # no arguments: every fixture; otherwise only the named generators (fixtures of earlier rounds stay untouched)
    for name in (sys.argv[1:] or list(GENERATORS)):
        GENERATORS[name]()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("%-28s %8.1f KB" % (f, os.path.getsize(os.path.join(HERE, f)) / 1024))
