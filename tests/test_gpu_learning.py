"""GPU tests of the continuous-learning helpers (csrc/sampling.hip, nfopp/learning.py) against the oracle, which
restates the same Philox stream: checks are exact up to libm rounding of log/cos/exp."""
import numpy as np
import pytest
import torch

from conftest import load_golden, max_abs

pytestmark = pytest.mark.gpu

gc = pytest.importorskip("gpu_common")
import nfopp  # noqa: E402
from oracle import nfopp_oracle as orc  # noqa: E402

F32 = np.float32


def test_device_checkers_vs_reference_labels():
    z = load_golden("g11_init_checkers.npz")
    poses = torch.tensor(z["poses"].astype(F32), device="cuda")
    rc = nfopp.DeviceRectangleChecker(z["car_obstacles"], (-0.3, 0.2, -0.3, 0.2), (0, 3, 0, 3))
    cd = nfopp.DeviceCircleChecker(z["corridor_obstacles"], 0.3, (0, 3, 0, 3))
    got_r = rc.labels(poses).cpu().numpy().astype(np.uint8)
    got_c = cd.labels(poses).cpu().numpy().astype(np.uint8)
    assert np.array_equal(got_c, z["circle_truth"])       # reference labels (float64 numpy) reproduced exactly
    assert np.array_equal(got_r, z["rect_truth"])
    assert np.array_equal(cd.labels(poses[:, :2].contiguous()).cpu().numpy().astype(np.uint8), z["circle_truth"])
    rng = np.random.default_rng(3)
    grid = (rng.uniform(size=(60, 80)) < 0.3).astype(np.uint8) * 255
    xy = rng.uniform(-1, 9, (5000, 2)).astype(F32)
    gchk = nfopp.DeviceGridChecker(grid, -0.5, 0.25, 0.1)
    got = gchk.labels(torch.tensor(xy, device="cuda")).cpu().numpy().astype(bool)
    assert np.array_equal(got, orc.grid_check(xy, grid, -0.5, 0.25, 0.1))
    assert 0.2 < got.mean() < 1.0
    assert cd.labels(torch.zeros(0, 3, device="cuda")).shape == (0,)


@pytest.mark.parametrize("D,tag", [(3, "a"), (2, "c")])
def test_batch_sampler_vs_oracle_and_shard_invariance(D, tag):
    z = load_golden("g1_onf.npz")
    onf, cfg = gc.make_onf(z[tag + "_cfg"], z[tag + "_params"])
    rng = np.random.default_rng(D)
    B, N, cap, nf = 5, 60, 40, 10
    prev = rng.uniform(0.2, 2.8, (B, N, D)).astype(F32)
    bounds = (-0.1, 3.1, -0.1, 3.1)
    sm = nfopp.BatchSampler(onf, B, N, 1.5, 0.02, 0.3, nf, cap, seed=21)
    prev_d = torch.tensor(prev, device="cuda")
    s1 = sm.draw(prev_d, bounds).cpu().numpy().reshape(B, -1, D)
    cand, age, smp = orc.sample_candidates(prev, None, None, cap, nf, 1.5, 0.02, 0.3, bounds, 21, 0)
    assert max_abs(s1[:, :N - 1], smp[:, :N - 1]) < 2e-5          # course (Box-Muller: log/cos rounding)
    assert max_abs(s1[:, N - 1 + cap:], smp[:, N - 1 + cap:]) < 1e-6
    assert max_abs(sm.cand.cpu().numpy()[:, :N - 1], cand) < 2e-6
    logits = sm.cand_out.cpu().numpy()[:, :N - 1, 0]
    pool, page, chosen = orc.resample_pool(cand, age, logits, cap, 21, 0)
    got_pool = sm.pool.cpu().numpy()
    # same candidates chosen (as sets: ordering inside the pool is irrelevant), same ages
    for b in range(B):
        assert max_abs(np.sort(got_pool[b].sum(1)), np.sort(pool[b].sum(1))) < 1e-5
    assert (sm.pool_age.cpu().numpy() == 1).all()
    assert max_abs(s1[:, N - 1:N - 1 + cap], got_pool) == 0
    # second draw: pool carried, ages grow
    s2 = sm.draw(prev_d, bounds)
    ages = sm.pool_age.cpu().numpy()
    assert set(np.unique(ages)) <= {1.0, 2.0} and (ages == 2).any()
    # shard invariance: trajectories 3..4 as their own shard give the same poses
    sm2 = nfopp.BatchSampler(onf, 2, N, 1.5, 0.02, 0.3, nf, cap, seed=21, traj_index_offset=3)
    t1 = sm2.draw(prev_d[3:].contiguous(), bounds).cpu().numpy().reshape(2, -1, D)
    assert np.array_equal(t1, s1[3:])


def test_batch_planner_learns_the_field_while_planning():
    """Continuous mode end to end on one GPU: the shared field's BCE loss falls while 64 trajectories are optimised."""
    torch.random.manual_seed(5)
    onf = nfopp.ONF(0, 1, use_cos=True, use_normal_init=True, bias=True, angle_encoding=True).to("cuda")
    rng = np.random.default_rng(0)
    obstacles = np.stack([np.full(10, 1.5), np.linspace(0.5, 2.5, 10)], 1)
    checker = nfopp.DeviceCircleChecker(obstacles, 0.3, (0, 3, 0, 3))
    B, N = 64, 128
    bounds = (-0.1, 3.1, -0.1, 3.1)
    starts = np.concatenate([rng.uniform(0.2, 0.8, (B, 1)), rng.uniform(0.3, 2.7, (B, 1)), rng.uniform(-1, 1, (B, 1))], 1)
    goals = np.concatenate([rng.uniform(2.2, 2.8, (B, 1)), rng.uniform(0.3, 2.7, (B, 1)), rng.uniform(-1, 1, (B, 1))], 1)
    hyper = nfopp.TrajectoryHyper(collision_weight=1, constraint_deltas_weight=20, multipliers_lr=0.1, bounds=bounds)
    planner = nfopp.BatchPlanner(onf, B, N, hyper, checker=checker, fit_lr=5e-2, angle_offset=0.3, seed=3)
    planner.init(starts.astype(F32), goals.astype(F32), bounds)
    losses = []
    for k in range(60):
        planner.step()
        losses.append(float(planner.fitter.last_loss))
    paths = planner.get_paths()
    assert np.isfinite(paths).all() and paths.shape == (B, N + 2, 3)
    assert np.mean(losses[-10:]) < 0.6 * np.mean(losses[:5])
    assert planner.sampler.pool_full and planner.fitter.step_count == 60
    # the learnt field separates free space from the wall
    probe = torch.tensor([[1.5, 1.5, 0.0], [0.5, 1.5, 0.0], [2.5, 0.4, 0.0]], device="cuda")
    logit = onf(probe).cpu().numpy()[:, 0]
    assert logit[0] > 0 > logit[1] and logit[2] < 0


@pytest.mark.parametrize("D", [3, 2])
def test_path_evaluation_and_early_stop_vs_oracle(D):
    z = load_golden("g1_onf.npz")
    onf, cfg = gc.make_onf(z["a_cfg" if D == 3 else "c_cfg"], z["a_params" if D == 3 else "c_params"])
    rng = np.random.default_rng(7 + D)
    B, N, sub = 9, 33, 5
    obstacles = np.array([[1.5, y] for y in np.linspace(0.0, 1.6, 9)])
    checker = nfopp.DeviceCircleChecker(obstacles, 0.25, (0, 3, 0, 3))
    starts = np.concatenate([rng.uniform(0.2, 0.6, (B, 1)), rng.uniform(0.3, 2.7, (B, 1)), rng.uniform(-3, 3, (B, 1))], 1)[:, :D].astype(F32)
    goals = np.concatenate([rng.uniform(2.4, 2.8, (B, 1)), rng.uniform(0.3, 2.7, (B, 1)), rng.uniform(-3, 3, (B, 1))], 1)[:, :D].astype(F32)
    hyper = nfopp.TrajectoryHyper(bounds=(0, 3, 0, 3)) if D == 3 else nfopp.TrajectoryHyper(collision_weight=0.01, bounds=(0, 3, 0, 3))
    planner = nfopp.BatchPlanner(onf, B, N, hyper)
    planner.init(starts, goals, (0, 3, 0, 3))
    eng = planner.engine
    traj0 = eng.traj.cpu().numpy().reshape(B, N, D)
    collides, length = planner.evaluate(checker, sub=sub, early_stop=True)
    torch.cuda.synchronize()
    poses, ref_len = orc.path_interpolate(traj0, starts, goals, sub)
    assert max_abs(planner._poses.cpu().numpy(), poses) < 1e-6
    assert max_abs(length.cpu().numpy(), ref_len) < 1e-5
    labels = orc.circle_check(poses.reshape(-1, D)[:, :2].astype(np.float64), obstacles, 0.25, (0, 3, 0, 3)).reshape(B, -1)
    ref_col, ref_best, ref_bl, ref_act = orc.path_select_best(labels, ref_len, traj0, traj0.copy(), np.full(B, np.inf, F32),
                                                              np.ones(B, bool))
    assert np.array_equal(collides.cpu().numpy().astype(bool), ref_col)
    assert 0 < ref_col.sum() < B                                   # straight lines: some cross the wall, some do not
    assert np.array_equal(np.isfinite(planner.best_length.cpu().numpy()), ~ref_col)
    assert eng.active.cpu().numpy().all()                          # first evaluation never retires (everything improves)
    # second evaluation of the SAME paths: collision-free ones do not improve -> retired; colliding ones stay active
    planner.evaluate(checker, sub=sub, early_stop=True)
    act = eng.active.cpu().numpy().astype(bool)
    assert np.array_equal(act, ref_col)
    # retired trajectories are frozen by the step kernels, active ones keep moving
    before = eng.traj.clone()
    for _ in range(11):
        planner.step()
    moved = (eng.traj - before).abs().reshape(B, -1).amax(1).cpu().numpy() > 0
    assert np.array_equal(moved, act)
    best = planner.best_paths()
    assert best.shape == (B, N + 2, D)
    assert max_abs(best[~ref_col][:, 1:-1], traj0[~ref_col]) == 0


def test_cell_indexed_circle_checker_equals_the_plain_one():
    """300 discs (bench.py's map): the cell-indexed kernel must give the labels of the all-pairs kernel, which the
    reference pins, on poses that include disc rims, map corners and far-away points."""
    rng = np.random.default_rng(1234)
    discs = rng.uniform(5, 95, (300, 2))
    bounds = (0.0, 100.0, 0.0, 100.0)
    fast = nfopp.DeviceCircleChecker(discs, 1.5, bounds)
    assert fast.cells is not None
    slow = nfopp.DeviceCircleChecker(discs, 1.5, bounds)
    slow.cells = None
    slow.obstacles = torch.tensor(discs.astype(F32), device="cuda")
    rng = np.random.default_rng(7)
    ang = rng.uniform(0, 2 * np.pi, 20000)
    rim = discs[rng.integers(0, 300, 20000)] + (1.5 + rng.choice([-1e-4, 0, 1e-4, -1e-6, 1e-6], 20000))[:, None] * np.stack([np.cos(ang), np.sin(ang)], 1)
    poses = np.concatenate([rng.uniform(-20, 120, (200000, 2)), rim, [[0, 0], [100, 100], [-1e6, 3], [50, 1e6]]]).astype(F32)
    poses = np.concatenate([poses, np.zeros((len(poses), 1), F32)], 1)
    p = torch.tensor(poses, device="cuda")
    a, b = fast.labels(p).cpu().numpy(), slow.labels(p).cpu().numpy()
    assert np.array_equal(a, b), (int((a != b).sum()), poses[a != b][:5])
    assert 0.1 < a.mean() < 0.9
    ref = orc.circle_check(poses[:, :2].astype(np.float64), discs.astype(F32).astype(np.float64), 1.5, bounds)
    bad = a.astype(bool) != ref                       # float64 reference: only poses within fp32 rounding of a rim may differ
    assert not bad[:200000].any() and not bad[-4:].any() and bad[200000:-4].mean() < 0.02
