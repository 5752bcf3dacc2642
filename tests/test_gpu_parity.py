"""GPU parity tests: every HIP kernel is driven through the C ABI (ctypes, libnfopp_hip.so) and compared with
(1) the committed golden vectors produced by the reference and (2) the CPU oracle on the same seeded inputs.

Tolerances (fp32): single op / single step 1e-5 relative (scaled to the array's max) -- BASELINE.md section 4;
K-step rollouts follow the reference-vs-itself drift: <=50 steps 2e-4, 200 steps 2e-2 (xy).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, max_abs, max_rel

pytestmark = pytest.mark.gpu

gc = pytest.importorskip("gpu_common")
import nfopp  # noqa: E402
from oracle import nfopp_oracle as orc  # noqa: E402

F32 = np.float32


def _eval_points(onf, x):
    out = onf.forward_with_grad(torch.tensor(x, device="cuda"))
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_onf_eval_points_vs_golden(tag):
    z = load_golden("g1_onf.npz")
    onf, cfg = gc.make_onf(z[tag + "_cfg"], z[tag + "_params"])
    x = z[tag + "_x"]
    out = _eval_points(onf, x)
    tol = 3e-5 if tag == "b" else 1e-5   # tag b: |e| ~ 40 rad, rounding of the encoding alone is ~5e-6
    assert gc.scaled_err(out[:, 0], z[tag + "_logit"]) < tol
    d = x.shape[1]
    assert gc.scaled_err(out[:, 1:1 + d], z[tag + "_grad"]) < 5 * tol
    if d == 2:
        assert np.all(out[:, 3] == 0)
    # and against the oracle on a ragged count (tail tile partially filled, several chunks)
    rng = np.random.default_rng(5)
    for n in (1, 17, 255, 4099):
        xs = x[rng.integers(0, len(x), n)]
        o = _eval_points(onf, xs)
        lo, go = orc.onf_forward_grad(z[tag + "_params"], cfg, xs)
        assert gc.scaled_err(o[:, 0], lo) < tol and gc.scaled_err(o[:, 1:1 + d], go) < 5 * tol


def test_onf_eval_empty_and_errors():
    z = load_golden("g1_onf.npz")
    onf, _ = gc.make_onf(z["a_cfg"], z["a_params"])
    assert onf.forward_with_grad(torch.zeros(0, 3, device="cuda")).shape == (0, 4)
    with pytest.raises(ValueError):
        onf.forward_with_grad(torch.zeros(4, 2, device="cuda"))


@pytest.mark.parametrize("name", ["traj_n100_default.npz", "traj_n100_hard.npz", "traj_n256_default.npz"])
def test_collision_eval_and_terms_vs_golden(name):
    z = load_golden(name)
    onf, cfg = gc.make_onf(z["cfg"], z["params"])
    hp = orc.Hyper.from_npz(z)
    s = gc.state_of(z, "s0_")
    eng = gc.engine_from_state(onf, s, hp)
    eng.collision_eval(z["g2_t"][None])
    torch.cuda.synchronize()
    out = eng.onf_out.cpu().numpy()[0]
    assert gc.scaled_err(out[:, 0], z["g2_logit"]) < 1e-5
    eng.update()
    terms = eng.loss_terms()
    for ours, ref in (("total", "total"), ("distance", "l_dist"), ("softplus_sum", "l_col"), ("cm_tanh", "l_cm"),
                      ("boundary", "l_bnd")):
        assert max_rel(terms[ours][0], z["g2_" + ref], 1e-4) < 2e-5, ours
    assert max_rel(terms["c_squared"][0], np.sum(z["g2_c"].astype(np.float64) ** 2), 1e-6) < 2e-5
    # lambda ascent exposes dL/dlambda = c exactly: lam_new - lam_old = lr * c
    lam_new = eng.lam.cpu().numpy()[0]
    assert max_abs((lam_new - s["lam"][0]) / hp.multipliers_lr, z["g2_c"]) < 2e-5


@pytest.mark.parametrize("name", ["traj_n100_default.npz", "traj_n100_hard.npz", "traj_n256_default.npz"])
def test_one_optimizer_step_vs_golden(name):
    z = load_golden(name)
    onf, cfg = gc.make_onf(z["cfg"], z["params"])
    hp = orc.Hyper.from_npz(z)
    s = gc.state_of(z, "s0_")
    eng = gc.engine_from_state(onf, s, hp)
    eng.optimize_trajectory(z["g3_t"][None])
    torch.cuda.synchronize()
    assert max_abs(eng.traj.cpu().numpy()[0], z["g3_traj"]) < 2e-6
    assert max_abs(eng.lam.cpu().numpy()[0], z["g3_lam"]) < 2e-6
    assert max_abs(eng.cm.cpu().numpy()[0], z["g3_cm"]) < 1e-6
    assert gc.scaled_err(eng.adam_m.cpu().numpy()[0], z["g3_adam_m"]) < 1e-5
    assert gc.scaled_err(eng.adam_v.cpu().numpy()[0], z["g3_adam_v"]) < 2e-5


@pytest.mark.parametrize("name,ks", [("traj_n100_default.npz", (1, 10, 50, 200)), ("traj_n100_hard.npz", (1, 10, 50)),
                                     ("traj_n256_default.npz", (1, 10))])
def test_rollouts_vs_golden(name, ks):
    z = load_golden(name)
    onf, cfg = gc.make_onf(z["cfg"], z["params"])
    hp = orc.Hyper.from_npz(z)
    s = gc.state_of(z, "g3_")
    eng = gc.engine_from_state(onf, s, hp)
    step_count = s["step_count"]
    tol = {1: 3e-6, 10: 2e-5, 50: 2e-4, 200: 2e-2}
    done = 0
    for K in ks:
        while done < K:
            eng.optimize_trajectory(z["g6_t"][done][None], want_terms=False)
            if step_count % 10 == 0:
                eng.reparametrize()
            step_count += 1
            done += 1
        pre = "g6_k%d_" % K
        tr = eng.traj.cpu().numpy()[0]
        assert step_count == int(z[pre + "step_count"])
        assert max_abs(tr[:, :2], z[pre + "traj"][:, :2]) < tol[K], K
        assert max_abs(tr[:, 2], z[pre + "traj"][:, 2]) < (10 if K >= 200 else 2) * tol[K], K
        assert max_abs(eng.lam.cpu().numpy()[0], z[pre + "lam"]) < 30 * tol[K], K
        assert max_abs(eng.cm.cpu().numpy()[0], z[pre + "cm"]) < tol[K], K


@pytest.mark.parametrize("tag", ["mid", "wrap", "clamp"])
def test_reparametrize_vs_golden(tag):
    z = load_golden("g4_reparam.npz")
    g1 = load_golden("g1_onf.npz")
    onf, _ = gc.make_onf(g1["a_cfg"], g1["a_params"])
    n = z[tag + "_in_traj"].shape[0]
    s = dict(traj=z[tag + "_in_traj"][None], start=z[tag + "_start"][None], goal=z[tag + "_goal"][None],
             lam=z[tag + "_in_lam"][None], cm=z[tag + "_in_cm"][None], adam_m=np.zeros((1, n, 3), F32),
             adam_v=np.zeros((1, n, 3), F32), adam_step=0)
    eng = gc.engine_from_state(onf, s, orc.Hyper())
    eng.reparametrize()
    torch.cuda.synchronize()
    # the cdf is built with torch-CPU's own roundings (norm / cascade sum / float64 cumsum: csrc/reparam.hip), so
    # searchsorted lands on the reference's indices in every case, the 1-ulp tie on the 20 duplicated waypoints of
    # "clamp" included; the lerps round every product and sum on its own like the reference's separate torch ops.
    # The whole operation is BIT-EXACT against the reference's outputs.
    assert np.array_equal(eng.traj.cpu().numpy()[0], z[tag + "_out_traj"])
    assert np.array_equal(eng.lam.cpu().numpy()[0], z[tag + "_out_lam"])
    assert np.array_equal(eng.cm.cpu().numpy()[0], z[tag + "_out_cm"])


@pytest.mark.parametrize("dim", [3, 2])
def test_collision_samples_are_the_reference_ones_bit_for_bit(dim):
    """The fused kernel forms its collision samples (constrained:78-81 / nerf:113-117) with the reference's roundings:
    evaluating the trajectory batch with injected t must equal, bit for bit, evaluating the oracle's sample poses
    through the same kernel in explicit-pose mode.  B*(N-1) = 130560 samples: the two-tiles-per-wave variant."""
    g1 = load_golden("g1_onf.npz")
    tag = "a" if dim == 3 else "c"
    onf, cfg = gc.make_onf(g1[tag + "_cfg"], g1[tag + "_params"])
    assert g1[tag + "_x"].shape[1] == dim
    rng = np.random.default_rng(5)
    B, N = 512, 256
    traj = (rng.random((B, N, dim)) * ([30.0, 30.0, 12.0][:dim]) - ([0.0, 0.0, 6.0][:dim])).astype(F32)
    t = rng.random((B, N - 1)).astype(F32)
    zeros = np.zeros((B, dim), F32)
    if dim == 3:
        s = dict(traj=traj, start=zeros, goal=zeros, lam=np.zeros((B, N + 1), F32), cm=np.zeros((B, N), F32),
                 adam_m=np.zeros((B, N, 3), F32), adam_v=np.zeros((B, N, 3), F32), adam_step=0)
        eng = gc.engine_from_state(onf, s, orc.Hyper())
        samples = orc.sample_collision_points(traj, t)
    else:
        hyper = nfopp.TrajectoryHyper(collision_weight=0.01, lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
        s = dict(traj=traj, start=zeros, goal=zeros, adam_m=np.zeros((B, N, 2), F32), adam_v=np.zeros((B, N, 2), F32),
                 adam_step=0)
        eng = gc.engine_from_state(onf, s, hyper, vh_weight=3.0)
        samples = orc.sample_collision_points_2d(traj, t)
    eng.collision_eval(t)
    torch.cuda.synchronize()
    from_traj = eng.onf_out.cpu().numpy().reshape(-1, 4)
    from_poses = onf.forward_with_grad(torch.tensor(samples.reshape(-1, dim), device="cuda")).cpu().numpy()
    assert np.array_equal(from_traj, from_poses)


def test_batch_equals_independent_reference_runs():
    z = load_golden("g8_batch.npz")
    onf, cfg = gc.make_onf(z["cfg"], z["params"])
    hp = orc.Hyper.from_npz(z)
    B, N = z["traj0"].shape[:2]
    s = dict(traj=z["traj0"].copy(), start=z["starts"], goal=z["goals"], lam=np.zeros((B, N + 1), F32),
             cm=np.zeros((B, N), F32), adam_m=np.zeros((B, N, 3), F32), adam_v=np.zeros((B, N, 3), F32), adam_step=0)
    eng = gc.engine_from_state(onf, s, hp)
    step_count = 1
    for k in range(int(z["steps"])):
        eng.optimize_trajectory(z["t"][:, k], want_terms=False)
        if step_count % 10 == 0:
            eng.reparametrize()
        step_count += 1
    # 12 steps incl. two reparametrisations, large heading changes: rounding differences of the field evaluation
    # (summation order of the hidden units) grow like the reference-vs-itself drift; gate = the 50-step level
    assert max_abs(eng.traj.cpu().numpy(), z["traj"]) < 2e-4
    assert max_abs(eng.lam.cpu().numpy(), z["lam"]) < 2e-3
    assert max_abs(eng.cm.cpu().numpy(), z["cm"]) < 2e-4


def test_planner_2d_vs_golden():
    z = load_golden("g10_planner2d.npz")
    onf, cfg = gc.make_onf(z["cfg"], z["params"])
    n = z["s0_traj"].shape[0]
    hyper = nfopp.TrajectoryHyper(collision_weight=float(z["collision_weight"]), lr=float(z["lr"]),
                                  betas=(float(z["beta1"]), float(z["beta2"])), eps=float(z["eps"]))
    s = dict(traj=z["s0_traj"][None], start=z["start"][None], goal=z["goal"][None], adam_m=z["s0_m"][None],
             adam_v=z["s0_v"][None], adam_step=int(z["s0_step"]))
    eng = gc.engine_from_state(onf, s, hyper, vh_weight=float(z["vh_weight"]))
    assert max_abs(eng.hinv, z["hinv"]) < 1e-7
    eng.optimize_trajectory(z["g3_t"][None])
    torch.cuda.synchronize()
    assert max_abs(eng.traj.cpu().numpy()[0], z["g3_traj"]) < 2e-6
    assert gc.scaled_err(eng.adam_m.cpu().numpy()[0], z["g3_m"]) < 1e-5
    eng.reparametrize()
    assert max_abs(eng.traj.cpu().numpy()[0], z["g4_traj"]) < 5e-6


def test_onf_training_step_vs_golden():
    z = load_golden("g7_onf_train.npz")
    onf, cfg = gc.make_onf(z["cfg"], z["params_before"])
    lib = nfopp.load_library()
    from nfopp import _lib
    x = torch.tensor(z["x"].astype(F32), device="cuda")
    y = torch.tensor(z["labels"].astype(F32), device="cuda")
    P = x.shape[0]
    c = onf.config_c()
    need = lib.nfopp_onf_train_workspace_bytes(c, P)
    ws = torch.empty((need + 3) // 4, dtype=torch.float32, device="cuda")
    grad = torch.zeros(onf.n_params + 2, device="cuda")
    _lib.check(lib.nfopp_onf_train_grad(c, _lib.ptr(onf.flat_parameters), _lib.ptr(x), _lib.ptr(y), P, 1.0 / P,
                                        _lib.ptr(grad), _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr()))
    torch.cuda.synchronize()
    g = grad.cpu().numpy()
    assert abs(float(g[-2]) - float(z["loss"])) < 2e-6
    assert g[-1] == P
    assert max_abs(g[:-2], z["grad"]) < 3e-6 * max(1.0, float(np.abs(z["grad"]).max()))
    # Adam on the flat buffer
    m = torch.tensor(z["adam_m_before"], device="cuda")
    v = torch.tensor(z["adam_v_before"], device="cuda")
    gref = torch.tensor(z["grad"], device="cuda")
    step = int(z["adam_step_before"]) + 1
    b1, b2, lr, eps = float(z["beta1"]), float(z["beta2"]), float(z["lr"]), float(z["eps"])
    _lib.check(lib.nfopp_adam_step(_lib.ptr(onf.flat_parameters), _lib.ptr(gref), _lib.ptr(m), _lib.ptr(v),
                                   onf.n_params, b2, 1 - b1, 1 - b2, eps, lr / (1 - b1 ** step),
                                   (1 - b2 ** step) ** 0.5, _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert max_abs(m.cpu().numpy(), z["adam_m_after"]) < 1e-7
    assert max_abs(v.cpu().numpy(), z["adam_v_after"]) < 1e-7
    assert max_abs(onf.flat_parameters.cpu().numpy(), z["params_after"]) < 2e-6


def test_device_philox_matches_numpy_and_shards_agree():
    z = load_golden("traj_n100_default.npz")
    onf, cfg = gc.make_onf(z["cfg"], z["params"])
    hp = orc.Hyper.from_npz(z)
    B = 6
    s = gc.state_of(z, "s0_", reps=B)
    s["traj"] = s["traj"] + np.linspace(0, 0.05, B, dtype=F32)[:, None, None]
    eng = gc.engine_from_state(onf, s, hp)
    eng.seed = 1234
    eng.collision_eval()
    t_dev = eng.t.cpu().numpy()
    N = s["traj"].shape[1]
    ref = gc.philox_uniform_np(1234, np.arange(B * (N - 1)), 0).reshape(B, N - 1)
    assert np.array_equal(t_dev, ref)
    assert t_dev.min() >= 0 and t_dev.max() < 1
    out_full = eng.onf_out.cpu().numpy()
    # a shard holding trajectories 4..5 with traj_index_offset = 4 must reproduce rows 4..5 bit for bit
    s2 = {k: (v[4:] if isinstance(v, np.ndarray) else v) for k, v in s.items()}
    eng2 = gc.engine_from_state(onf, s2, hp)
    eng2.seed, eng2.traj_index_offset = 1234, 4
    eng2.collision_eval()
    assert np.array_equal(eng2.t.cpu().numpy(), t_dev[4:])
    assert np.array_equal(eng2.onf_out.cpu().numpy(), out_full[4:])
    # second draw uses the next counter word
    eng.collision_eval()
    assert np.array_equal(eng.t.cpu().numpy(), gc.philox_uniform_np(1234, np.arange(B * (N - 1)), 1).reshape(B, N - 1))


def test_full_size_batch_replicates_single_trajectory():
    """BASELINE config-3 shape (4096 x 256): 4096 copies of one golden state with the same injected t must all
    equal the reference's single-trajectory result (size-independent property: trajectories are independent)."""
    z = load_golden("traj_n256_default.npz")
    onf, cfg = gc.make_onf(z["cfg"], z["params"])
    hp = orc.Hyper.from_npz(z)
    B = 4096
    s = gc.state_of(z, "s0_", reps=B)
    eng = gc.engine_from_state(onf, s, hp)
    eng.optimize_trajectory(np.repeat(z["g3_t"][None], B, axis=0))
    torch.cuda.synchronize()
    tr = eng.traj.cpu().numpy()
    assert np.array_equal(tr, np.repeat(tr[:1], B, axis=0))          # every row bit-identical
    assert max_abs(tr[0], z["g3_traj"]) < 2e-6                          # and equal to the reference's step
    assert max_abs(eng.lam.cpu().numpy()[B - 1], z["g3_lam"]) < 2e-6
    eng.reparametrize()
    tr2 = eng.traj.cpu().numpy()
    assert np.array_equal(tr2, np.repeat(tr2[:1], B, axis=0))


def _train_grad(onf, x, y, path, inv_count=None):
    lib = nfopp.load_library()
    from nfopp import _lib
    P = x.shape[0]
    c = onf.config_c()
    need = lib.nfopp_onf_train_workspace_bytes(c, P)
    ws = torch.empty((need + 3) // 4, dtype=torch.float32, device="cuda")
    grad = torch.zeros(onf.n_params + 2, device="cuda")
    _lib.check(lib.nfopp_onf_train_grad_ex(c, _lib.ptr(onf.flat_parameters), _lib.ptr(x), _lib.ptr(y), P,
                                           1.0 / P if inv_count is None else inv_count, _lib.ptr(grad), _lib.ptr(ws),
                                           ws.numel() * 4, path, _lib.stream_ptr()))
    torch.cuda.synchronize()
    return grad.cpu().numpy()


@pytest.mark.parametrize("path", [1, 2])
def test_onf_training_paths_vs_golden(path):
    """Both implementations of the fitting-step gradient (per-sample kernels / MFMA GEMM path) against the reference."""
    z = load_golden("g7_onf_train.npz")
    onf, cfg = gc.make_onf(z["cfg"], z["params_before"])
    x = torch.tensor(z["x"].astype(F32), device="cuda")
    y = torch.tensor(z["labels"].astype(F32), device="cuda")
    g = _train_grad(onf, x, y, path)
    assert abs(float(g[-2]) - float(z["loss"])) < 2e-6
    assert g[-1] == x.shape[0]
    assert max_abs(g[:-2], z["grad"]) < 3e-6 * max(1.0, float(np.abs(z["grad"]).max()))


@pytest.mark.parametrize("tag,P", [("a", 5000), ("b", 3001), ("c", 2500), ("a", 70001)])   # 70001: two point tiles per wave
def test_onf_training_mfma_path_vs_oracle_large(tag, P):
    """Ragged sample counts through the automatic (MFMA) path, three field configurations, vs the oracle."""
    z = load_golden("g1_onf.npz")
    onf, cfg = gc.make_onf(z[tag + "_cfg"], z[tag + "_params"])
    rng = np.random.default_rng(P)
    x = z[tag + "_x"][rng.integers(0, len(z[tag + "_x"]), P)].copy()
    x += rng.normal(0, 0.05, x.shape).astype(F32)
    y = (rng.uniform(size=P) < 0.4).astype(F32)
    g_auto = _train_grad(onf, torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda"), 0)
    loss, _, gref = orc.onf_train_grads(z[tag + "_params"], cfg, x, y)
    scale = max(1.0, float(np.abs(gref).max()))
    assert abs(float(g_auto[-2]) - float(loss)) < 5e-6 * max(1.0, abs(float(loss)))
    assert max_abs(g_auto[:-2], gref) < 2e-5 * scale
    if P <= 65536:      # the per-sample path (small fits) takes at most 65536 samples
        g_ps = _train_grad(onf, torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda"), 1)
        assert max_abs(g_ps[:-2], gref) < 2e-5 * scale
        # the two device paths differ only by summation order
        assert max_abs(g_auto[:-2], g_ps[:-2]) < 2e-5 * scale
    # bitwise reproducible
    g_again = _train_grad(onf, torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda"), 0)
    assert np.array_equal(g_auto, g_again)


@pytest.mark.parametrize("tag", ["a", "c"])
def test_forward_only_kernel_matches_the_full_kernel(tag):
    z = load_golden("g1_onf.npz")
    onf, cfg = gc.make_onf(z[tag + "_cfg"], z[tag + "_params"])
    for n in (3, 1024, 70001):
        x = torch.tensor(np.resize(z[tag + "_x"], (n, z[tag + "_x"].shape[1])), device="cuda")
        full = onf.forward_with_grad(x)
        fwd = onf(x)
        assert fwd.shape == (n, 1)
        assert torch.equal(fwd[:, 0], full[:, 0])          # same arithmetic, bit for bit
    assert gc.scaled_err(onf(torch.tensor(z[tag + "_x"], device="cuda")).cpu().numpy()[:, 0], z[tag + "_logit"]) < 1e-5
