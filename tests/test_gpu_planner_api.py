"""GPU tests of the drop-in boundary: PlannerFactory / planner classes driven exactly like the reference's drivers
(scripts/benchmark.py:22-101) and compared with what the reference produced from the same seeds."""
import numpy as np
import pytest
import torch

from conftest import abs_percentile, load_golden, max_abs

pytestmark = pytest.mark.gpu

import nfopp  # noqa: E402


def _params(n=100):
    A = nfopp.AttributeDict
    return A(device="cuda", trajectory_length=n,
             collision_model=A(mean=0, sigma=1, use_cos=True, bias=True, use_normal_init=True, angle_encoding=True, name="ONF"),
             trajectory_initializer=A(name="TrajectoryInitializer", resolution=0.05),
             collision_optimizer=A(lr=5e-2, betas=(0.9, 0.9)), trajectory_optimizer=A(lr=1e-2, betas=(0.9, 0.9)),
             planner=A(name="ConstrainedNERFOptPlanner", trajectory_random_offset=0.02, collision_weight=1,
                       velocity_hessian_weight=0.5, random_field_points=10, init_collision_iteration=0,
                       constraint_deltas_weight=20, multipliers_lr=0.1, init_collision_points=100,
                       reparametrize_trajectory_freq=10, optimize_collision_model_freq=1, angle_weight=0.5,
                       angle_offset=0.3, boundary_weight=1, collision_multipliers_lr=1e-3))


def _make(z, n=100):
    torch.random.manual_seed(100)
    np.random.seed(400)
    cc = nfopp.CircleDirectedCollisionChecker(0.3, (0, 3, 0, 3))
    cc.update_obstacle_points(z["obstacles"])
    cc.update_boundaries(tuple(z["bounds"]))
    planner = nfopp.PlannerFactory.make_constrained_onf_planner(cc, _params(n))
    planner.init(z["start"], z["goal"], tuple(z["bounds"]))
    return planner


def test_full_steps_with_onf_learning_follow_the_reference():
    """Same seeds, same driver calls: sampled poses, labels, field weights and the trajectory track the reference.
    ONF learning is on, so differences grow with the step count: gates widen per step."""
    z = load_golden("g9_full_steps.npz")
    planner = _make(z)
    assert isinstance(planner, nfopp.ConstrainedNERFOptPlanner)
    assert np.array_equal(planner._collision_model.flat_parameters.cpu().numpy(), z["params0"])
    assert max_abs(planner._trajectory.detach().cpu().numpy(), z["traj0"]) < 1e-6
    for k in range(int(z["steps"])):
        planner.step()
        checked = planner.checked_positions.as_vec()
        assert checked.shape == z["k%d_checked" % k].shape                    # 109 + 99 poses in steady state
        tol = 2e-6 * 4 ** k
        assert max_abs(checked[:99], z["k%d_checked" % k][:99]) < max(tol, 1e-5)   # course samples: same numpy draws
        assert max_abs(checked[-10:], z["k%d_checked" % k][-10:]) < 1e-12            # uniform field samples
        same_pool = max_abs(checked, z["k%d_checked" % k]) < 1e-4
        if same_pool:   # retained-pool resampling is a weighted np.random.choice: identical unless a draw ties
            assert np.array_equal(np.asarray(planner.truth_collision).astype(np.uint8), z["k%d_truth" % k])
        assert max_abs(planner._trajectory.detach().cpu().numpy(), z["k%d_traj" % k]) < 5e-6 * 4 ** k
        assert max_abs(planner._collision_model.flat_parameters.cpu().numpy(), z["k%d_params" % k]) < 2e-5 * 4 ** k
        assert max_abs(planner._constraint_multipliers.cpu().numpy(), z["k%d_lam" % k]) < 2e-5 * 4 ** k
    path = planner.get_path()
    assert path.shape == (102, 3) and path.dtype == np.float32
    assert np.array_equal(path[0], z["start"]) and np.array_equal(path[-1], z["goal"])
    assert planner._step_count == int(z["steps"])
    assert float(planner.last_onf_loss) > 0


def test_run_planner_script_configuration_follows_the_reference():
    """BASELINE configs[0] as scripts/run_planner.py:10-66 runs it: car environment, the off-centre rectangle robot
    (-0.3, 0.2, -0.3, 0.2) with checker bounds (0, 3, 0, 3) and ONLY update_obstacle_points called on the checker,
    seeds 100 / 400, six full `.step()`s with ONF learning (fixture g18, made by running the reference)."""
    z = load_golden("g18_run_planner_script.npz")
    torch.random.manual_seed(100)
    np.random.seed(400)
    cc = nfopp.RectangleCollisionChecker(tuple(z["box"]), tuple(z["checker_bounds"]))
    cc.update_obstacle_points(z["obstacles"])
    planner = nfopp.PlannerFactory.make_constrained_onf_planner(cc, _params(100))
    planner.init(z["start"], z["goal"], tuple(z["bounds"]))
    assert np.array_equal(planner._collision_model.flat_parameters.cpu().numpy(), z["params0"])
    assert max_abs(planner._trajectory.detach().cpu().numpy(), z["traj0"]) < 1e-6
    for k in range(int(z["steps"])):
        planner.step()
        checked = planner.checked_positions.as_vec()
        assert checked.shape == z["k%d_checked" % k].shape
        tol = 2e-6 * 4 ** k
        assert max_abs(checked[:99], z["k%d_checked" % k][:99]) < max(tol, 1e-5)
        assert max_abs(checked[-10:], z["k%d_checked" % k][-10:]) < 1e-12
        if max_abs(checked, z["k%d_checked" % k]) < 1e-4:   # same retained pool (see the g9 test): identical labels
            assert np.array_equal(np.asarray(planner.truth_collision).astype(np.uint8), z["k%d_truth" % k])
        assert max_abs(planner._trajectory.detach().cpu().numpy(), z["k%d_traj" % k]) < 5e-6 * 4 ** k
        if ("k%d_params" % k) in z.files:
            # Adam's first steps: an entry whose gradient is ~eps moves by lr * g / (|g| + eps) (lr 5e-2), so a rounding-level
            # change of g shows in single entries (the g15 test has the account): the bulk tightly, the maximum loosely
            got = planner._collision_model.flat_parameters.cpu().numpy()
            assert abs_percentile(got, z["k%d_params" % k], 99) < 2e-7 * 4 ** k
            assert max_abs(got, z["k%d_params" % k]) < 2e-3
        assert max_abs(planner._constraint_multipliers.cpu().numpy(), z["k%d_lam" % k]) < 2e-5 * 4 ** k
        assert max_abs(planner._collision_multipliers.cpu().numpy(), z["k%d_cm" % k]) < 2e-5 * 4 ** k


def test_update_goal_and_start_point_vs_golden():
    z9 = load_golden("g9_full_steps.npz")
    z = load_golden("g4_update_endpoints.npz")
    planner = _make(z9)
    eng = planner._engine
    eng.traj.copy_(torch.tensor(z["in_traj"]))
    eng.lam.copy_(torch.tensor(z["in_lam"][None]))
    eng.cm.copy_(torch.tensor(z["in_cm"][None]))
    eng.set_endpoints(z["start"][None], z["goal"][None])
    planner._step_count = 7
    planner.update_goal_point(z["new_goal"])
    assert planner._step_count == 0
    assert max_abs(planner._trajectory.detach().cpu().numpy(), z["goal_out_traj"]) < 5e-6
    assert max_abs(planner._collision_multipliers.cpu().numpy(), z["goal_out_cm"]) < 5e-6
    assert max_abs(planner._constraint_multipliers.cpu().numpy(), z["goal_out_lam"]) < 5e-6
    assert np.array_equal(planner._goal_point.cpu().numpy()[0], z["new_goal"])
    planner.update_start_point(z["new_start"])
    assert max_abs(planner._trajectory.detach().cpu().numpy(), z["start_out_traj"]) < 5e-6
    assert max_abs(planner._collision_multipliers.cpu().numpy(), z["start_out_cm"]) < 5e-6
    assert max_abs(planner._constraint_multipliers.cpu().numpy(), z["start_out_lam"]) < 5e-6
    planner.set_boundaries((0, 1, 0, 1))
    assert planner._random_sample_border == (0, 1, 0, 1) and planner._step_count == 0


def test_2d_planner_factory_shapes_and_steps():
    """The reference's own unit tests for this class pin shapes and endpoint copies (test/test_nerf_opt_planner.py:28-50)."""
    torch.random.manual_seed(100)
    np.random.seed(400)
    z = load_golden("g10_planner2d.npz")
    g11 = load_golden("g11_init_checkers.npz")
    cc = nfopp.CircleCollisionChecker(0.3, (0, 3, 0, 3))
    cc.update_obstacle_points(g11["corridor_obstacles"])
    planner = nfopp.PlannerFactory.make_onf_planner(cc)
    planner._init_collision_iteration = 40
    start, goal = np.array([0.5, 0.5], np.float32), np.array([2.5, 2.5], np.float32)
    planner.init(start, goal, (-0.1, 3.1, -0.1, 3.1))
    assert np.array_equal(planner.get_path()[0], start) and np.array_equal(planner.get_path()[-1], goal)
    assert planner.get_path().shape == (102, 2) and tuple(planner.full_trajectory().shape) == (102, 2)
    for _ in range(15):
        planner.step()
    tr = planner._trajectory.detach().cpu().numpy()
    assert np.isfinite(tr).all()
    # same seeds and call order as the fixture run (40 fits + 15 steps with ONF learning): stays close
    assert max_abs(tr, z["s0_traj"]) < 2e-3
    terms = planner.trajectory_loss_terms()
    assert terms["total"] > 0 and terms["lambda_dot_c"] == 0


def test_factory_rejects_cpu_device():
    p = _params()
    p.device = "cpu"
    with pytest.raises(RuntimeError, match="MI355X"):
        nfopp.PlannerFactory.make_constrained_onf_planner(nfopp.CollisionChecker(), p)
