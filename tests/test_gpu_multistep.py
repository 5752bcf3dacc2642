"""GPU: `step(n)` -- n planner steps enqueued by one library call (nfopp_traj_steps, ABI 6) -- equals n single steps bit for
bit: batch planner (device Philox and injected draws, incl. the reparametrisation schedule and the per-step Adam scalars
formed on the C side), the drop-in planner with a frozen field and with ONF fitting steps falling inside the run, and the
torch.ops form.  The single steps themselves are pinned to the reference's fixtures in the other GPU test modules."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

gc = pytest.importorskip("gpu_common")
import nfopp  # noqa: E402
from nfopp import torch_ops  # noqa: E402
from oracle import nfopp_oracle as orc  # noqa: E402
from test_gpu_planner_api import _make  # noqa: E402

F32 = np.float32


def _batch(onf, hp, B, N, seed=5, freq=10):
    rng = np.random.default_rng(11)
    starts = np.concatenate([rng.uniform(0.3, 0.9, (B, 2)), rng.uniform(-3, 3, (B, 1))], 1).astype(F32)
    goals = np.concatenate([rng.uniform(2.1, 2.7, (B, 2)), rng.uniform(-3, 3, (B, 1))], 1).astype(F32)
    p = nfopp.BatchPlanner(onf, B, N, hp, device="cuda", seed=seed, reparametrize_trajectory_freq=freq)
    p.init(starts, goals, hp.bounds)
    return p


def _state(p):
    e = p.engine
    torch.cuda.synchronize()
    return [x.cpu().numpy().copy() for x in (e.traj, e.lam, e.cm, e.adam_m, e.adam_v, e.terms)] + [e.adam_step, e.rng_offset,
                                                                                                      p.step_count]


def _same(a, b):
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("n,freq", [(1, 10), (25, 10), (37, 7)])
def test_batch_step_n_equals_n_steps_bit_for_bit(n, freq):
    z = load_golden("traj_n100_hard.npz")
    onf, cfg = gc.make_onf(z["cfg"], z["params"])
    hp = gc.hyper_from(orc.Hyper.from_npz(z))
    one, many = _batch(onf, hp, 3, 64, freq=freq), _batch(onf, hp, 3, 64, freq=freq)
    for k in range(n):
        one.engine.collision_eval()                      # the single-step sequence, spelled out
        one.engine.update(k == n - 1)
        if one.step_count % one.reparam_freq == 0:
            one.engine.reparametrize()
        one.step_count += 1
    many.step(n=n, want_terms=True)
    _same(_state(one), _state(many))
    # and a second call continues the schedule (Adam step count, Philox word, reparametrisation phase)
    for _ in range(5):
        one.step(want_terms=True)
    many.step(n=5, want_terms=True)
    _same(_state(one), _state(many))
    assert np.isfinite(many.get_paths()).all()


def test_batch_step_n_with_injected_draws():
    z = load_golden("traj_n100_default.npz")
    onf, cfg = gc.make_onf(z["cfg"], z["params"])
    hp = gc.hyper_from(orc.Hyper.from_npz(z))
    B, N, n = 2, 100, 12
    t = np.random.default_rng(2).uniform(0, 1, (n, B, N - 1)).astype(F32)
    one, many = _batch(onf, hp, B, N), _batch(onf, hp, B, N)
    for k in range(n):
        one.step(t[k], want_terms=True)
    many.step(t, want_terms=True, n=n)
    _same(_state(one), _state(many))


@pytest.mark.parametrize("fit_freq", [10 ** 9, 4])
def test_drop_in_step_n_equals_n_steps_bit_for_bit(fit_freq):
    """ConstrainedNERFOptPlanner.step(n): the reference's generator order is kept (torch draws t per step; the numpy draws of
    an ONF fitting step that falls inside the run come where they came before)."""
    z = load_golden("g9_full_steps.npz")
    res = []
    for chunked in (False, True):
        planner = _make(z)
        planner.step()                                   # step 0 fits the field once in both runs
        planner._optimize_collision_model_freq = fit_freq
        if chunked:
            planner.step(7)
            planner.step(16)
        else:
            for _ in range(23):
                planner.step()
        path = planner.get_path()
        res.append((path, planner._constraint_multipliers.cpu().numpy(), planner._collision_multipliers.cpu().numpy(),
                    planner._collision_model.flat_parameters.cpu().numpy(), planner._step_count,
                    np.asarray(list(planner.trajectory_loss_terms().values()))))
    for a, b in zip(*res):
        assert np.array_equal(a, b)
    assert res[0][4] == 24 and not planner._collision_model.is_frozen


def test_traj_steps_torch_op_equals_the_ctypes_path():
    ops = torch_ops.load()
    z = load_golden("traj_n100_hard.npz")
    onf, cfg = gc.make_onf(z["cfg"], z["params"])
    hp = gc.hyper_from(orc.Hyper.from_npz(z))
    a, b = _batch(onf, hp, 2, 64), _batch(onf, hp, 2, 64)
    a.step(n=13, want_terms=True)
    e = b.engine
    ca = (float(cfg.mean), float(cfg.sigma), bool(cfg.use_cos), bool(cfg.bias), 10 if cfg.angle_encoding else 0)
    ops.traj_steps(onf.flat_parameters, *ca, e.traj, e.start, e.goal, e.lam, e.cm, e.adam_m, e.adam_v, e.t, None, e.seed,
                   e.rng_offset, e.traj_index_offset, e.onf_out, e.hinv_band, e.half_width, e.interior[0], e.interior[1], e.u,
                   torch_ops.hyper_list(hp.to_c(1)), float(hp.lr), float(hp.betas[0]), float(hp.betas[1]), e.adam_step,
                   b.step_count, b.reparam_freq, 13, e.terms, None, None)
    e.adam_step += 13
    e.rng_offset += 13
    b.step_count += 13
    _same(_state(a), _state(b))
    with pytest.raises(RuntimeError, match="t_steps"):
        ops.traj_steps(onf.flat_parameters, *ca, e.traj, e.start, e.goal, e.lam, e.cm, e.adam_m, e.adam_v, e.t,
                       torch.zeros(3, 2, 63, device="cuda"), e.seed, e.rng_offset, e.traj_index_offset, e.onf_out, e.hinv_band,
                       e.half_width, e.interior[0], e.interior[1], e.u, torch_ops.hyper_list(hp.to_c(1)), float(hp.lr),
                       float(hp.betas[0]), float(hp.betas[1]), e.adam_step, b.step_count, b.reparam_freq, 13, None, None, None)


def test_step_n_honours_the_active_mask():
    """Early stop inside `step(n)`: retired trajectories are compacted out of the ONF kernel and skipped by the update and the
    reparametrisation in every one of the n steps -- their state stays bit for bit, the live ones equal n single steps."""
    z = load_golden("traj_n100_hard.npz")
    onf, cfg = gc.make_onf(z["cfg"], z["params"])
    hp = gc.hyper_from(orc.Hyper.from_npz(z))
    one, many = _batch(onf, hp, 5, 64), _batch(onf, hp, 5, 64)
    for p in (one, many):
        p.step(n=3)
        p.engine.active = torch.tensor([1, 0, 1, 0, 1], dtype=torch.uint8, device="cuda")
    frozen = [x.clone() for x in (many.engine.traj, many.engine.lam, many.engine.cm, many.engine.adam_m)]
    for _ in range(12):
        one.step()
    many.step(n=12)
    _same(_state(one), _state(many))
    for before, after in zip(frozen, (many.engine.traj, many.engine.lam, many.engine.cm, many.engine.adam_m)):
        assert torch.equal(before[1], after[1]) and torch.equal(before[3], after[3])
        assert not torch.equal(before[0], after[0])


def test_step_n_at_the_benchmark_size_equals_single_steps():
    """BASELINE configs[2] shape (4096 x 256, the 512-thread workgroup shape of the fused kernel): 12 steps from one call equal
    12 single steps bit for bit, reparametrisations included."""
    z = load_golden("traj_benchmr_n256.npz")
    onf, cfg = gc.make_onf(z["cfg"], z["params"])
    hp = gc.hyper_from(orc.Hyper.from_npz(z))
    B, N = 4096, 256
    rng = np.random.default_rng(3)
    lo, hi = hp.bounds[0] + 2, hp.bounds[1] - 2
    starts = np.concatenate([rng.uniform(lo, hi, (B, 2)), rng.uniform(-3, 3, (B, 1))], 1).astype(F32)
    goals = np.concatenate([rng.uniform(lo, hi, (B, 2)), rng.uniform(-3, 3, (B, 1))], 1).astype(F32)
    res = []
    for chunked in (False, True):
        p = nfopp.BatchPlanner(onf, B, N, hp, device="cuda", seed=100)
        p.init(starts, goals, hp.bounds)
        if chunked:
            p.step(n=12)
        else:
            for _ in range(12):
                p.engine.collision_eval()
                p.engine.update(False)
                if p.step_count % p.reparam_freq == 0:
                    p.engine.reparametrize()
                p.step_count += 1
        torch.cuda.synchronize()
        res.append([x.clone() for x in (p.engine.traj, p.engine.lam, p.engine.cm, p.engine.adam_m, p.engine.adam_v)])
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert torch.isfinite(res[0][0]).all()
