"""Pin the CPU oracle (oracle/nfopp_oracle.py) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import BENCHMR_FIXTURES, benchmr_rollout_tol, check_benchmr_rollout, check_batch_snapshot, load_golden, max_abs, max_rel
from oracle import nfopp_oracle as orc

F32 = np.float32


def scaled_err(a, b):
    """max |a-b| relative to the array's scale (gradients mix large and tiny entries)."""
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(np.asarray(a, np.float64) - b)) / (np.max(np.abs(b)) + 1e-12))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g1_onf_logit_and_input_grad(tag):
    z = load_golden("g1_onf.npz")
    cfg = orc.OnfConfig.from_vector(z[tag + "_cfg"])
    logit, grad = orc.onf_forward_grad(z[tag + "_params"], cfg, z[tag + "_x"])
    # sigma=10 field (tag b): |e| reaches ~40 rad, fp32 rounding of e alone is ~5e-6
    tol = 3e-5 if tag == "b" else 1e-5
    assert scaled_err(logit, z[tag + "_logit"]) < tol
    assert scaled_err(grad, z[tag + "_grad"]) < 5 * tol


def _state(z, prefix):
    keys = ("traj", "start", "goal", "lam", "cm", "adam_m", "adam_v")
    s = {k: z[prefix + k][None].astype(F32) for k in keys}
    s["adam_step"] = int(z[prefix + "adam_step"])
    s["step_count"] = int(z[prefix + "step_count"])
    return s


@pytest.mark.parametrize("name", ["traj_n100_default.npz", "traj_n100_hard.npz", "traj_n256_default.npz"])
def test_g2_loss_terms_and_grads(name):
    z = load_golden(name)
    cfg = orc.OnfConfig.from_vector(z["cfg"])
    hp = orc.Hyper.from_npz(z)
    s = _state(z, "s0_")
    t = z["g2_t"][None]
    pts = orc.sample_collision_points(s["traj"], t)
    assert max_abs(pts[0], z["g2_pos"]) < 1e-6
    logit, dl = orc.onf_forward_grad(z["params"], cfg, pts[0])
    assert scaled_err(logit, z["g2_logit"]) < 1e-5
    terms = orc.trajectory_loss_terms(s["traj"], s["start"], s["goal"], s["lam"], s["cm"], t, logit[None], dl[None], hp)
    for k in ("total", "l_dist", "l_col", "l_cm", "l_bnd"):
        assert max_rel(terms[k][0], z["g2_" + k], 1e-4) < 2e-5, k
    assert max_abs(terms["c"][0], z["g2_c"]) < 1e-6
    assert max_abs(terms["d"][0], z["g2_d"]) < 1e-6
    assert scaled_err(terms["g_traj"][0], z["g2_g_traj"]) < 1e-5
    assert max_abs(terms["g_lam"][0], z["g2_g_lam"]) < 1e-6
    assert max_abs(terms["g_cm"][0], z["g2_g_cm"]) < 2e-6
    if "hard" in name:  # the fixture must actually exercise the rare branches
        assert (z["g2_d"] > 0).sum() > 5 and float(z["g2_l_bnd"]) > 0


@pytest.mark.parametrize("name", ["traj_n100_default.npz", "traj_n100_hard.npz", "traj_n256_default.npz"])
def test_g3_one_optimizer_step(name):
    z = load_golden(name)
    cfg = orc.OnfConfig.from_vector(z["cfg"])
    hp = orc.Hyper.from_npz(z)
    s = _state(z, "s0_")
    tr, lam, cm, m, v, _ = orc.optimize_trajectory(s["traj"], s["start"], s["goal"], s["lam"], s["cm"], s["adam_m"],
                                                   s["adam_v"], s["adam_step"], z["g3_t"][None], z["params"], cfg, hp,
                                                   z["hinv"])
    assert max_abs(tr[0], z["g3_traj"]) < 2e-6
    assert max_abs(lam[0], z["g3_lam"]) < 1e-6
    assert max_abs(cm[0], z["g3_cm"]) < 1e-6
    assert scaled_err(m[0], z["g3_adam_m"]) < 1e-5
    assert scaled_err(v[0], z["g3_adam_v"]) < 2e-5


@pytest.mark.parametrize("name,ks", [("traj_n100_default.npz", (1, 10, 50, 200)), ("traj_n100_hard.npz", (1, 10, 50)),
                                     ("traj_n256_default.npz", (1, 10))])
def test_g6_rollouts(name, ks):
    """Per-horizon tolerances follow the reference-vs-itself drift measured in SURVEY 8(c)."""
    z = load_golden(name)
    cfg = orc.OnfConfig.from_vector(z["cfg"])
    hp = orc.Hyper.from_npz(z)
    s = _state(z, "g3_")
    tol = {1: 3e-6, 10: 2e-5, 50: 2e-4, 200: 2e-2}
    done = 0
    for K in ks:
        while done < K:
            orc.planner_step(s, z["g6_t"][done][None], z["params"], cfg, hp, z["hinv"])
            done += 1
        pre = "g6_k%d_" % K
        assert s["step_count"] == int(z[pre + "step_count"])
        # xy and theta separately: theta is the chaotic component (reference vs itself, 1 vs 8 threads:
        # 1.5e-3 / 1.2e-3 @200 steps, 1.8e-2 / 0.16 @500 -- SURVEY 8(c)); at K=200 theta gets 10x the xy gate
        assert max_abs(s["traj"][0][:, :2], z[pre + "traj"][:, :2]) < tol[K], K
        assert max_abs(s["traj"][0][:, 2], z[pre + "traj"][:, 2]) < (10 if K >= 200 else 2) * tol[K], K
        assert max_abs(s["lam"][0], z[pre + "lam"]) < 30 * tol[K], K
        assert max_abs(s["cm"][0], z[pre + "cm"]) < tol[K], K


@pytest.mark.parametrize("tag", ["mid", "wrap", "clamp"])
def test_g4_reparametrize(tag):
    z = load_golden("g4_reparam.npz")
    tr, lam, cm = orc.reparametrize(z[tag + "_in_traj"][None], z[tag + "_start"][None], z[tag + "_goal"][None],
                                    z[tag + "_in_lam"][None], z[tag + "_in_cm"][None])
    # the cdf is restated with torch's own roundings (norm / cascade sum / float64 cumsum), so searchsorted lands on
    # the reference's indices in every case -- also on the 20 duplicated waypoints of "clamp", where the grid value
    # ties with the cdf to 1 ulp -- and what is left is the rounding of the final lerps
    tol = 2e-6
    assert max_abs(tr[0], z[tag + "_out_traj"]) < tol
    assert max_abs(lam[0], z[tag + "_out_lam"]) < tol
    assert max_abs(cm[0], z[tag + "_out_cm"]) < tol


def test_torch_reduction_orders():
    """The three torch-CPU roundings behind the reparametrisation cdf, against torch itself, bit for bit."""
    import torch
    rng = np.random.default_rng(0)
    for n in list(range(1, 140)) + [255, 256, 257, 258, 511, 513, 514, 1025, 2049, 4100]:
        a = (rng.uniform(0, 1, n) ** 3).astype(F32)
        assert float(torch.sum(torch.tensor(a))) == float(orc.torch_sum_f32(a)), n
    d = rng.normal(0, 1, (50000, 2)).astype(F32)
    assert np.array_equal(torch.norm(torch.tensor(d), dim=1).numpy(), orc.torch_norm2_f32(d[:, 0], d[:, 1]))
    a = rng.uniform(0, 1, 513).astype(F32)
    a /= a.sum()
    assert np.array_equal(torch.cumsum(torch.tensor(a), 0).numpy(), np.cumsum(a.astype(np.float64)).astype(F32))


def test_g5_inverse_hessian():
    z = load_golden("g5_hinv.npz")
    for n in (16, 100):
        for w, ws in ((0.5, "0p5"), (3.0, "3p0")):
            assert max_abs(orc.calculate_inv_hessian(n, w), z["n%d_w%s" % (n, ws)]) < 1e-7
    for n in (256, 512):
        for w, ws in ((0.5, "0p5"), (3.0, "3p0")):
            h = orc.calculate_inv_hessian(n, w)
            band = z["n%d_w%s_band64" % (n, ws)]
            for i in (0, 1, n // 2, n - 1):
                lo, hi = max(0, i - 64), min(n, i + 65)
                assert max_abs(h[i, lo:hi], band[i, lo - i + 64:hi - i + 64]) < 1e-7


def test_g7_onf_training_step():
    z = load_golden("g7_onf_train.npz")
    cfg = orc.OnfConfig.from_vector(z["cfg"])
    loss, logit, grad = orc.onf_train_grads(z["params_before"], cfg, z["x"].astype(F32), z["labels"])
    assert abs(float(loss) - float(z["loss"])) < 1e-6
    assert scaled_err(logit, z["logit"]) < 1e-5
    assert max_abs(grad, z["grad"]) < 2e-6 * max(1.0, float(np.abs(z["grad"]).max()))
    step = int(z["adam_step_before"]) + 1
    p, m, v = orc.adam_update(z["params_before"], z["grad"], z["adam_m_before"], z["adam_v_before"], step,
                              float(z["lr"]), float(z["beta1"]), float(z["beta2"]), float(z["eps"]))
    assert max_abs(m, z["adam_m_after"]) < 1e-7
    assert max_abs(v, z["adam_v_after"]) < 1e-7
    assert max_abs(p, z["params_after"]) < 2e-6


@pytest.mark.parametrize("name,ks", BENCHMR_FIXTURES)
def test_benchmr_settings_terms_step_rollouts(name, ks):
    """The settings bench.py runs (bench-mr hyper block, fitted sigma=10 field, 100 m disc map, N = 256 / 512): loss
    terms + gradients, one optimiser step and frozen-field rollouts against the reference (G2/G3/G6 form)."""
    z = load_golden(name)
    cfg = orc.OnfConfig.from_vector(z["cfg"])
    hp = orc.Hyper.from_npz(z)
    assert (hp.collision_weight, hp.collision_beta, hp.direction_delta_weight, hp.angle_weight, hp.lr) == (100, 10, 100, 5, 5e-2)
    n = z["s0_traj"].shape[0]
    hinv = orc.calculate_inv_hessian(n, float(z["vh_weight"]))   # pinned by g5 (N = 256 and 512)
    s = _state(z, "s0_")
    t = z["g2_t"][None]
    pts = orc.sample_collision_points(s["traj"], t)
    assert max_abs(pts[0], z["g2_pos"]) < 1e-5                   # coordinates up to 100: 1 ulp = 7.6e-6
    logit, dl = orc.onf_forward_grad(z["params"], cfg, pts[0])
    assert scaled_err(logit, z["g2_logit"]) < 1e-5
    terms = orc.trajectory_loss_terms(s["traj"], s["start"], s["goal"], s["lam"], s["cm"], t, logit[None], dl[None], hp)
    for k in ("total", "l_dist", "l_col", "l_cm", "l_bnd"):
        assert max_rel(terms[k][0], z["g2_" + k], 1e-4) < 2e-5, k
    assert max_abs(terms["c"][0], z["g2_c"]) < 1e-6
    assert max_abs(terms["d"][0], z["g2_d"]) < 1e-6
    assert scaled_err(terms["g_traj"][0], z["g2_g_traj"]) < 2e-5
    assert max_abs(terms["g_cm"][0], z["g2_g_cm"]) < 1e-5
    if "n256" in name:   # the fixture exercises the linear softplus branch (beta * logit > 20) and the direction term
        assert (10 * z["g2_logit"] > 20).sum() >= 1 and (z["g2_d"] > 0).sum() > 50
    tr, lam, cm, m, v, _ = orc.optimize_trajectory(s["traj"], s["start"], s["goal"], s["lam"], s["cm"], s["adam_m"],
                                                   s["adam_v"], s["adam_step"], z["g3_t"][None], z["params"], cfg, hp, hinv)
    assert max_abs(tr[0], z["g3_traj"]) < 1e-5
    assert max_abs(lam[0], z["g3_lam"]) < 1e-6
    assert max_abs(cm[0], z["g3_cm"]) < 1e-6
    assert scaled_err(m[0], z["g3_adam_m"]) < 1e-5
    assert scaled_err(v[0], z["g3_adam_v"]) < 2e-5
    s = _state(z, "g3_")
    done = 0
    for K in ks:
        while done < K:
            orc.planner_step(s, z["g6_t"][done][None], z["params"], cfg, hp, hinv)
            done += 1
        assert s["step_count"] == int(z["g6_k%d_step_count" % K])
        check_benchmr_rollout(name, K, s["traj"][0], s["lam"][0], s["cm"][0], z)


def test_benchmr_small_batch_equals_independent_runs():
    """B = 4 reference problems on the benchmarked settings (N = 256, 12 steps from the straight-line start, two
    reparametrisations), snapshots after steps 1 / 3 / 12 (gates: conftest.BENCHMR_BATCH_TOL)."""
    z = load_golden("g14_benchmr_batch.npz")
    cfg = orc.OnfConfig.from_vector(z["cfg"])
    hp = orc.Hyper.from_npz(z)
    B, N = z["traj0"].shape[:2]
    hinv = orc.calculate_inv_hessian(N, 0.5)
    s = dict(traj=z["traj0"].copy(), start=z["starts"], goal=z["goals"], lam=np.zeros((B, N + 1), F32),
             cm=np.zeros((B, N), F32), adam_m=np.zeros((B, N, 3), F32), adam_v=np.zeros((B, N, 3), F32),
             adam_step=0, step_count=1)
    for k in range(int(z["steps"])):
        orc.planner_step(s, z["t"][:, k], z["params"], cfg, hp, hinv)
        if k + 1 in z["snapshots"]:
            check_batch_snapshot(k + 1, s["traj"], s["lam"], s["cm"], z)


def test_g16_grid_checker_labels():
    """MapCollisionChecker labels made by the notebook's own class (cell 2 exec'd by make_golden.py g16)."""
    z = load_golden("g16_grid_checker.npz")
    assert z["grid"].shape == (100, 100) and 0.2 < (z["grid"] == 0).mean() < 0.5
    for tag in ("unit", "fine"):
        ox, oy, cell = (float(v) for v in z[tag + "_geom"])
        got = orc.grid_check(z[tag + "_poses"][:, :2], z["grid"], ox, oy, cell)
        assert np.array_equal(got.astype(np.uint8), z[tag + "_truth"])
        assert 0.3 < z[tag + "_truth"].mean() < 0.9


def test_g8_batch_equals_independent_runs():
    z = load_golden("g8_batch.npz")
    cfg = orc.OnfConfig.from_vector(z["cfg"])
    hp = orc.Hyper.from_npz(z)
    B, N = z["traj0"].shape[:2]
    s = dict(traj=z["traj0"].copy(), start=z["starts"], goal=z["goals"], lam=np.zeros((B, N + 1), F32),
             cm=np.zeros((B, N), F32), adam_m=np.zeros((B, N, 3), F32), adam_v=np.zeros((B, N, 3), F32),
             adam_step=0, step_count=1)
    for k in range(int(z["steps"])):
        orc.planner_step(s, z["t"][:, k], z["params"], cfg, hp, z["hinv"])
    assert max_abs(s["traj"], z["traj"]) < 3e-5
    assert max_abs(s["lam"], z["lam"]) < 3e-4
    assert max_abs(s["cm"], z["cm"]) < 3e-5


def test_g10_planner_2d():
    z = load_golden("g10_planner2d.npz")
    cfg = orc.OnfConfig.from_vector(z["cfg"])
    tr0 = z["s0_traj"][None]
    start, goal = z["start"][None], z["goal"][None]
    t = z["g2_t"][None]
    pts = orc.sample_collision_points_2d(tr0, t)
    logit, dl = orc.onf_forward_grad(z["params"], cfg, pts[0])
    terms = orc.trajectory_loss_2d(tr0, start, goal, t, logit[None], dl[None], float(z["collision_weight"]))
    assert max_rel(terms["total"][0], z["g2_total"]) < 1e-5
    assert scaled_err(terms["g_traj"][0], z["g2_grad"]) < 1e-5
    tr, m, v, _ = orc.optimize_trajectory_2d(tr0, start, goal, z["s0_m"][None], z["s0_v"][None], int(z["s0_step"]),
                                             z["g3_t"][None], z["params"], cfg, float(z["collision_weight"]),
                                             float(z["lr"]), float(z["beta1"]), float(z["beta2"]), float(z["eps"]),
                                             z["hinv"])
    assert max_abs(tr[0], z["g3_traj"]) < 2e-6
    assert scaled_err(m[0], z["g3_m"]) < 1e-5
    out = orc.reparametrize(z["g3_traj"][None], start, goal)
    assert max_abs(out[0], z["g4_traj"]) < 5e-6


def test_g11_initializer_and_checkers():
    z = load_golden("g11_init_checkers.npz")
    for c, ref in zip(z["init_cases"], z["init_traj"]):
        assert max_abs(orc.initialize_trajectory(c[:3], c[3:], 50), ref) < 1e-6
    poses = z["poses"]
    rect = orc.rectangle_check(poses, z["car_obstacles"], (-0.3, 0.2, -0.3, 0.2), (0, 3, 0, 3))
    assert np.array_equal(rect.astype(np.uint8), z["rect_truth"])
    circ = orc.circle_check(poses[:, :2], z["corridor_obstacles"], 0.3, (0, 3, 0, 3))
    assert np.array_equal(circ.astype(np.uint8), z["circle_truth"])


def test_linspace_matches_torch():
    import torch
    for a, b, n in ((0.0, 1.0, 102), (0.5, 2.5, 258), (-3.0, 2.9, 52), (0.0, 1.0, 514)):
        assert np.array_equal(orc.linspace_f32(a, b, n), torch.linspace(a, b, n).numpy())



# Note on the one known answer printed in the reference's files (notebooks/pytorch-optimal-planner.ipynb cell 8:
# 0.0728476345539093): that prototype starts Adam from gradients that are exactly zero up to rounding noise (1e-9), and
# Adam's first step turns rounding noise into +-lr moves, so the value is only reproducible with bit-identical op
# order (torch 2.10 here: 0.07284770; this oracle's arithmetic: 0.0719).  It also uses an earlier form of the
# non-holonomic term (start angle instead of the mean angle).  It therefore cannot pin a restatement; the fixtures
# generated from the shipped planner classes (tests/golden/make_golden.py) do.
