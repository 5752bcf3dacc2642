"""GPU edge cases against the oracle: minimum / ragged / long trajectories, wide preconditioner bands, empty batch,
2-D fields, both step kernels and the reparametrisation on random (not optimised) states."""
import numpy as np
import pytest
import torch

from conftest import load_golden, max_abs

pytestmark = pytest.mark.gpu

gc = pytest.importorskip("gpu_common")
import nfopp  # noqa: E402
from oracle import nfopp_oracle as orc  # noqa: E402

F32 = np.float32


def _random_state(rng, B, N, D, bounds):
    lo, hi = bounds[0], bounds[1]
    tr = rng.uniform(lo - 0.2, hi + 0.2, (B, N, D)).astype(F32)
    st = rng.uniform(lo, hi, (B, D)).astype(F32)
    go = rng.uniform(lo, hi, (B, D)).astype(F32)
    if D == 3:
        tr[..., 2] = rng.uniform(-3.5, 3.5, (B, N))
        st[:, 2] = rng.uniform(-3.1, 3.1, B)
        go[:, 2] = rng.uniform(-3.1, 3.1, B)
    s = dict(traj=tr, start=st, goal=go, adam_m=(rng.normal(size=(B, N, D)) * 0.1).astype(F32),
             adam_v=(rng.uniform(0, 0.05, (B, N, D))).astype(F32), adam_step=int(rng.integers(0, 40)), step_count=3)
    if D == 3:
        s["lam"] = (rng.normal(size=(B, N + 1)) * 0.2).astype(F32)
        s["cm"] = rng.uniform(0, 0.1, (B, N)).astype(F32)
    return s


# the last three: velocity-Hessian weights whose boundary band columns exceed K2's LDS budget (csrc/traj_update.hip:
# 205 KB at N = 512, w = 10; no Toeplitz interior at all at N = 256, w = 10) -- those waypoints read the band from global memory
@pytest.mark.parametrize("B,N,w", [(1, 2, 0.5), (2, 3, 0.5), (3, 17, 0.5), (5, 100, 3.0), (3, 512, 0.5), (2, 700, 3.0),
                                   (2, 512, 10.0), (2, 256, 10.0), (1, 512, 20.0)])
def test_se2_step_and_reparam_vs_oracle(B, N, w):
    z = load_golden("g1_onf.npz")
    onf, cfg = gc.make_onf(z["a_cfg"], z["a_params"])
    bounds = (-0.1, 3.1, -0.1, 3.1)
    hp = orc.Hyper(collision_weight=3, angle_weight=0.7, constraint_deltas_weight=15, multipliers_lr=0.05,
                   collision_multipliers_lr=2e-3, boundary_weight=2, collision_beta=2.5, direction_delta_weight=6,
                   lr=2e-2, beta1=0.9, beta2=0.95, bounds=bounds)
    rng = np.random.default_rng(N * 7 + B)
    s = _random_state(rng, B, N, 3, bounds)
    eng = gc.engine_from_state(onf, s, hp, vh_weight=w)
    t = rng.uniform(0, 1, (B, N - 1)).astype(F32)
    hinv = orc.calculate_inv_hessian(N, w)
    assert max_abs(eng.hinv, hinv) < 1e-7
    eng.optimize_trajectory(t)
    tr, lam, cm, m, v, terms = orc.optimize_trajectory(s["traj"], s["start"], s["goal"], s["lam"], s["cm"], s["adam_m"],
                                                       s["adam_v"], s["adam_step"], t, z["a_params"], cfg, hp, hinv)
    torch.cuda.synchronize()
    # random states have O(1e2) gradients through the constraint terms; compare relative to the step scale
    scale = max(1.0, float(np.abs(terms["g_traj"]).max()))
    assert max_abs(eng.traj.cpu().numpy(), tr) < 6e-6
    assert max_abs(eng.lam.cpu().numpy(), lam) < 3e-6 * max(1.0, float(np.abs(terms["c"]).max()))
    assert max_abs(eng.cm.cpu().numpy(), cm) < 1e-6
    assert gc.scaled_err(eng.adam_m.cpu().numpy(), m) < 2e-5
    assert gc.scaled_err(eng.adam_v.cpu().numpy(), v) < 4e-5
    got = eng.loss_terms()
    for ours, ref in (("total", "total"), ("distance", "l_dist"), ("softplus_sum", "l_col"), ("lambda_dot_c", "l_lin"),
                      ("c_squared", "l_c2"), ("boundary", "l_bnd"), ("cm_tanh", "l_cm"), ("direction", "l_dir")):
        assert np.allclose(got[ours], terms[ref], rtol=3e-5, atol=3e-5 * scale), ours
    eng.reparametrize()
    rtr, rlam, rcm = orc.reparametrize(tr, s["start"], s["goal"], lam, cm)
    torch.cuda.synchronize()
    # random zig-zag paths have arbitrarily short segments: tau = (u - cdf_b) / (cdf_a - cdf_b) amplifies the 1-ulp
    # difference in the cdf normalisation by 1 / segment-fraction (the well-conditioned cases are pinned by the
    # golden fixtures in test_gpu_parity.py); positions move by at most |q_a - q_b| * d tau
    assert max_abs(eng.traj.cpu().numpy(), rtr) < 1e-3
    assert max_abs(eng.cm.cpu().numpy(), rcm) < 1e-3
    assert max_abs(eng.lam.cpu().numpy(), rlam) < 1e-2


@pytest.mark.parametrize("B,N", [(1, 2), (4, 37), (2, 300)])
def test_2d_step_and_reparam_vs_oracle(B, N):
    z = load_golden("g1_onf.npz")
    onf, cfg = gc.make_onf(z["c_cfg"], z["c_params"])
    rng = np.random.default_rng(N)
    s = _random_state(rng, B, N, 2, (0, 3))
    hyper = nfopp.TrajectoryHyper(collision_weight=0.3, lr=1e-2, betas=(0.9, 0.999))
    eng = gc.engine_from_state(onf, s, hyper, vh_weight=3.0)
    t = rng.uniform(0, 1, (B, N - 1)).astype(F32)
    hinv = orc.calculate_inv_hessian(N, 3.0)
    eng.optimize_trajectory(t)
    tr, m, v, terms = orc.optimize_trajectory_2d(s["traj"], s["start"], s["goal"], s["adam_m"], s["adam_v"], s["adam_step"],
                                                 t, z["c_params"], cfg, 0.3, 1e-2, 0.9, 0.999, 1e-8, hinv)
    torch.cuda.synchronize()
    assert max_abs(eng.traj.cpu().numpy(), tr) < 3e-6
    assert gc.scaled_err(eng.adam_m.cpu().numpy(), m) < 2e-5
    got = eng.loss_terms()
    assert np.allclose(got["total"], terms["total"], rtol=3e-5)
    eng.reparametrize()
    assert max_abs(eng.traj.cpu().numpy(), orc.reparametrize(tr, s["start"], s["goal"])) < 2e-4


def test_empty_batch_and_argument_errors():
    z = load_golden("g1_onf.npz")
    onf, _ = gc.make_onf(z["a_cfg"], z["a_params"])
    eng = nfopp.TrajectoryEngine(onf, 0, 16, 3, nfopp.TrajectoryHyper(), 0.5, "cuda")
    eng.optimize_trajectory(np.zeros((0, 15), F32))
    eng.reparametrize()
    torch.cuda.synchronize()
    onf2, _ = gc.make_onf(z["c_cfg"], z["c_params"])
    with pytest.raises(ValueError):
        nfopp.TrajectoryEngine(onf2, 1, 16, 3, nfopp.TrajectoryHyper(), 0.5, "cuda")   # 2-D field, SE(2) trajectory
    lib = nfopp.load_library()
    from nfopp import _lib
    c = onf.config_c()
    x = torch.zeros(4, 3, device="cuda")
    assert lib.nfopp_traj_collision_eval(c, _lib.ptr(onf.flat_parameters), _lib.ptr(x), 1, 1, 3, _lib.ptr(x), 0, 0, 0, 0,
                                         _lib.ptr(x), None, None, None) == -1           # fewer than 2 waypoints
    assert lib.nfopp_traj_collision_eval(c, _lib.ptr(onf.flat_parameters), _lib.ptr(x), 1, 4, 3, _lib.ptr(x), 7, 0, 0, 0,
                                         _lib.ptr(x), None, None, None) == -1           # bad t_mode
    mask = torch.ones(1, dtype=torch.uint8, device="cuda")
    assert lib.nfopp_traj_collision_eval(c, _lib.ptr(onf.flat_parameters), _lib.ptr(x), 1, 4, 3, _lib.ptr(x), 0, 0, 0, 0,
                                         _lib.ptr(x), _lib.ptr(mask, torch.uint8), None, None) == -1   # mask without workspace
    with pytest.raises(nfopp.NfoppError):
        _lib.ptr(torch.zeros(3))                                                         # host tensor where a device one is due
