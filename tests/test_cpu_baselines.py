"""The two CPU baselines bench.py times beside the GPU number (oracle/cpu_baselines.py) against the fixtures the
reference produced: they must BE the reference's step, or their timings mean nothing.  CPU only."""
import numpy as np
import pytest
import torch

from conftest import benchmr_rollout_tol, load_golden, max_abs, max_rel
from oracle import cpu_baselines as cb
from oracle import nfopp_oracle as orc

F32 = np.float32
CASES = [("traj_benchmr_n256.npz", 10), ("traj_n100_hard.npz", 10)]


def _setup(name):
    z = load_golden(name)
    cfg = z["cfg"]
    field = cb.Field(z["params"], cfg[0], cfg[1])
    sc = cb.Scalars.from_oracle(orc.Hyper.from_npz(z), velocity_hessian_weight=float(z["vh_weight"]))
    return z, field, sc


def _state(z, pre):
    return dict(traj=z[pre + "traj"], start=z[pre + "start"], goal=z[pre + "goal"], lam=z[pre + "lam"], cm=z[pre + "cm"],
                adam_m=z[pre + "adam_m"], adam_v=z[pre + "adam_v"], adam_step=int(z[pre + "adam_step"]),
                step_count=int(z[pre + "step_count"]))


def _scaled(a, b):
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(np.asarray(a, np.float64) - b)) / (np.max(np.abs(b)) + 1e-12))


def _rollout_tol(name, K):
    if "benchmr" in name:
        return benchmr_rollout_tol(name, K)
    return dict(xy=2e-5, th=4e-5, lam=6e-4, cm=2e-5)     # the K = 10 gates of tests/test_oracle_golden.py::test_g6_rollouts


@pytest.mark.parametrize("name,K", CASES)
def test_batched_autograd_free_baseline_is_the_reference_step(name, K):
    z, field, sc = _setup(name)
    s = _state(z, "s0_")
    pl = cb.BatchedTorchPlanner(field, sc, s["traj"][None], s["start"][None], s["goal"][None], s["lam"][None], s["cm"][None],
                                s["adam_m"][None], s["adam_v"][None], s["adam_step"], s["step_count"])
    pl.optimize_trajectory(z["g3_t"][None])
    assert max_abs(pl.traj[0].numpy(), z["g3_traj"]) < 1e-5
    assert max_abs(pl.lam[0].numpy(), z["g3_lam"]) < 2e-6
    assert max_abs(pl.cm[0].numpy(), z["g3_cm"]) < 1e-6
    assert _scaled(pl.m[0].numpy(), z["g3_adam_m"]) < 1e-5
    # loss terms of the G2 draw from the same state
    pl = cb.BatchedTorchPlanner(field, sc, s["traj"][None], s["start"][None], s["goal"][None], s["lam"][None], s["cm"][None],
                                s["adam_m"][None], s["adam_v"][None], s["adam_step"], s["step_count"])
    pl.optimize_trajectory(z["g2_t"][None])
    for ours, ref in (("total", "total"), ("l_dist", "l_dist"), ("l_col", "l_col"), ("l_cm", "l_cm"), ("l_bnd", "l_bnd")):
        assert max_rel(pl.terms[ours][0].numpy(), z["g2_" + ref], 1e-4) < 2e-5, ours
    assert max_abs(pl.terms["c"][0].numpy(), z["g2_c"]) < 1e-6
    # K frozen-field steps incl. reparametrisation, two trajectories at once (batch axis)
    s = _state(z, "g3_")
    two = lambda a: np.stack([a, a])  # noqa: E731
    pl = cb.BatchedTorchPlanner(field, sc, two(s["traj"]), two(s["start"]), two(s["goal"]), two(s["lam"]), two(s["cm"]),
                                two(s["adam_m"]), two(s["adam_v"]), s["adam_step"], s["step_count"])
    for k in range(K):
        pl.step(two(z["g6_t"][k]))
    tol, pre = _rollout_tol(name, K), "g6_k%d_" % K
    for b in range(2):
        assert max_abs(pl.traj[b, :, :2].numpy(), z[pre + "traj"][:, :2]) < tol["xy"]
        assert max_abs(pl.traj[b, :, 2].numpy(), z[pre + "traj"][:, 2]) < tol["th"]
        assert max_abs(pl.lam[b].numpy(), z[pre + "lam"]) < tol["lam"]
        assert max_abs(pl.cm[b].numpy(), z[pre + "cm"]) < tol["cm"]


@pytest.mark.parametrize("name,K", CASES)
def test_eager_autograd_baseline_is_the_reference_step(name, K):
    torch.set_num_threads(1)
    z, field, sc = _setup(name)
    s = _state(z, "s0_")
    pl = cb.EagerAutogradPlanner(field, sc, **s)
    total = pl.loss(z["g2_t"])
    total.backward()
    assert max_rel(float(total.detach()), z["g2_total"], 1e-4) < 2e-5
    assert _scaled(pl.traj.grad.numpy(), z["g2_g_traj"]) < 2e-5
    assert max_abs(pl.lam.grad.numpy(), z["g2_g_lam"]) < 1e-6
    assert max_abs(pl.cm.grad.numpy(), z["g2_g_cm"]) < 1e-5
    pl = cb.EagerAutogradPlanner(field, sc, **s)
    pl.optimize_trajectory(z["g3_t"])
    assert max_abs(pl.traj.detach().numpy(), z["g3_traj"]) < 1e-5
    assert max_abs(pl.lam.detach().numpy(), z["g3_lam"]) < 2e-6
    assert max_abs(pl.cm.detach().numpy(), z["g3_cm"]) < 1e-6
    st = pl.optimizer.state[pl.traj]
    assert _scaled(st["exp_avg"].numpy(), z["g3_adam_m"]) < 1e-5 and float(st["step"]) == float(z["g3_adam_step"])
    pl = cb.EagerAutogradPlanner(field, sc, **_state(z, "g3_"))
    for k in range(K):
        pl.step(z["g6_t"][k])
    tol, pre = _rollout_tol(name, K), "g6_k%d_" % K
    assert max_abs(pl.traj.detach().numpy()[:, :2], z[pre + "traj"][:, :2]) < tol["xy"]
    assert max_abs(pl.traj.detach().numpy()[:, 2], z[pre + "traj"][:, 2]) < tol["th"]
    assert max_abs(pl.lam.detach().numpy(), z[pre + "lam"]) < tol["lam"]
    assert max_abs(pl.cm.detach().numpy(), z[pre + "cm"]) < tol["cm"]


def test_timing_harness_runs_on_a_tiny_sample(tmp_path):
    z, field, sc = _setup("traj_benchmr_n256.npz")
    rng = np.random.default_rng(0)
    starts = np.concatenate([rng.uniform(5, 95, (4, 2)), rng.uniform(-3, 3, (4, 1))], 1).astype(F32)
    goals = np.concatenate([rng.uniform(5, 95, (4, 2)), rng.uniform(-3, 3, (4, 1))], 1).astype(F32)
    kw = {("sc_" + k): np.asarray(v, np.float64) for k, v in sc.__dict__.items()}
    np.savez(tmp_path / "in.npz", onf_flat=z["params"], onf_cfg=z["cfg"], n_waypoints=32, starts=starts, goals=goals, **kw)
    out = cb.time_baselines(str(tmp_path / "in.npz"), 0.3, 0.4, 2)
    assert out["strong"]["value"] > 0 and out["strong"]["kind"] == "port"
    assert out["reference_faithful"]["value"] > 0 and out["reference_faithful"]["kind"] == "reference-faithful"
    assert out["reference_faithful"]["process_parallel"]["procs"] == 2
    one = cb.time_baselines(str(tmp_path / "in.npz"), 0.2, 0.2, 1)      # what bench.py asks for on a GPU box
    assert "process_parallel" not in one["reference_faithful"]
