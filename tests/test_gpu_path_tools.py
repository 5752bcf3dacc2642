"""GPU: device trajectory initialiser (csrc/traj_init.hip) and path post-processor (csrc/path_post.hip) through the
C ABI, against outputs of the reference itself (tests/golden/g11, g12) and the oracle on random batches."""
import numpy as np
import pytest
import torch

from conftest import load_golden, max_abs

pytestmark = pytest.mark.gpu

gc = pytest.importorskip("gpu_common")
import nfopp  # noqa: E402
from oracle import nfopp_oracle as orc  # noqa: E402

F32 = np.float32


def _dev(x):
    return torch.tensor(np.ascontiguousarray(x, dtype=F32), device="cuda")


def test_initialiser_vs_reference():
    z = load_golden("g11_init_checkers.npz")
    c = z["init_cases"]
    got = nfopp.init_trajectories(_dev(c[:, :3]), _dev(c[:, 3:]), 50).cpu().numpy()
    assert np.array_equal(got, z["init_traj"])                  # torch.linspace rounding reproduced bit for bit
    z = load_golden("g12_init_dir_postprocess.npz")
    c = z["dir_cases"]
    for n in (50, 51):
        got = nfopp.init_trajectories(_dev(c[:, :3]), _dev(c[:, 3:]), n, init_angles_with_trajectory=True).cpu().numpy()
        want = z["dir_traj_n%d" % n]
        assert np.array_equal(got[..., :2], want[..., :2])
        assert max_abs(got[..., 2], want[..., 2]) < 1e-6        # atan2f rounding
    # 2-D point robots, ragged sizes, empty batch
    rng = np.random.default_rng(5)
    s, g = rng.uniform(-3, 3, (7, 2)).astype(F32), rng.uniform(-3, 3, (7, 2)).astype(F32)
    got = nfopp.init_trajectories(_dev(s), _dev(g), 1).cpu().numpy()
    for b in range(7):
        for k in range(2):
            assert got[b, 0, k] == orc.linspace_f32(s[b, k], g[b, k], 3)[1]
    assert nfopp.init_trajectories(torch.zeros(0, 3, device="cuda"), torch.zeros(0, 3, device="cuda"), 9).shape == (0, 9, 3)
    with pytest.raises(nfopp.NfoppError):
        nfopp.init_trajectories(_dev(s), _dev(g), 10, init_angles_with_trajectory=True)   # needs headings


def test_initialiser_batch_vs_oracle_and_planner_init():
    rng = np.random.default_rng(6)
    B, N = 33, 257
    s = np.concatenate([rng.uniform(0, 100, (B, 2)), rng.uniform(-3.1, 3.1, (B, 1))], 1).astype(F32)
    g = np.concatenate([rng.uniform(0, 100, (B, 2)), rng.uniform(-3.1, 3.1, (B, 1))], 1).astype(F32)
    got = nfopp.init_trajectories(_dev(s), _dev(g), N).cpu().numpy()
    want = np.stack([orc.initialize_trajectory(s[b], g[b], N) for b in range(B)])
    assert np.array_equal(got[..., :2], want[..., :2])
    assert max_abs(got[..., 2], want[..., 2]) < 5e-7            # wrap of the heading difference (fmod vs fma form)
    assert np.array_equal(got, nfopp.straight_line_init(s, g, N)) or max_abs(got, nfopp.straight_line_init(s, g, N)) < 5e-7
    got = nfopp.init_trajectories(_dev(s), _dev(g), N, init_angles_with_trajectory=True).cpu().numpy()
    want = np.stack([orc.initialize_trajectory_directed(s[b], g[b], N) for b in range(B)])
    assert max_abs(got, want) < 2e-6
    # BatchPlanner.init uses the device initialiser
    z = load_golden("g1_onf.npz")
    onf, _ = gc.make_onf(z["a_cfg"], z["a_params"])
    bp = nfopp.BatchPlanner(onf, B, N, nfopp.TrajectoryHyper(), init_angles_with_trajectory=True)
    bp.init(s, g, (0, 100, 0, 100))
    assert max_abs(bp.engine.traj.cpu().numpy(), want) < 2e-6


def test_postprocessor_vs_reference():
    z = load_golden("g12_init_dir_postprocess.npz")
    pp = nfopp.PathPostprocessor()
    for i in range(5):
        res = pp.process(nfopp.Position2.from_vec(z["post_in_%d" % i].copy())).as_vec()
        want = z["post_out_%d" % i]
        assert res.shape == want.shape and res.dtype == np.float64
        assert max_abs(res, want) < 1e-11
    alt = nfopp.PathPostprocessor(*z["post_alt_params"])
    res = alt.process(nfopp.Position2.from_vec(z["post_in_1"].copy())).as_vec()
    assert res.shape == z["post_out_alt_1"].shape and max_abs(res, z["post_out_alt_1"]) < 1e-11
    short = nfopp.Position2.from_vec(z["post_in_0"][:2].copy())
    assert pp.process(short) is short                           # < 3 poses pass through unchanged (:14-15)
    with pytest.raises(ValueError):
        pp.process(nfopp.Position2.from_vec(np.zeros((5, 3), F32)))


def test_postprocessor_batch_vs_oracle():
    rng = np.random.default_rng(8)
    B, n = 40, 258
    s = np.linspace(0, 1, n)[None, :]
    amp, ph = rng.uniform(1, 20, (B, 1)), rng.uniform(0, 6, (B, 1))
    x = 40 * s * rng.uniform(0.2, 1, (B, 1)) + 3
    y = amp * np.sin(3 * s + ph)
    th = np.arctan2(np.gradient(y, axis=1), np.gradient(x, axis=1)) + rng.normal(0, 0.05, (B, n))
    paths = np.stack([x, y, th], 2).astype(F32)
    paths[3, 100:140] = paths[3, 100]                           # a parked stretch: filtered out
    paths[5, :4, 0] += np.asarray([0.5, 0.3, 0.2, 0.1], F32)    # reversing start
    out, counts = nfopp.PathPostprocessor().process_batch(paths)
    out, counts = out.cpu().numpy(), counts.cpu().numpy()
    for b in range(B):
        want = orc.path_postprocess(paths[b])
        assert counts[b] == len(want), b
        assert max_abs(out[b, :counts[b]], want) < 1e-10, b
    assert counts.max() == out.shape[1]
    # long paths (n = 1026 poses is the kernel's limit) and size errors
    n = 1026
    s = np.linspace(0, 1, n)
    long = np.stack([30 * s, 5 * np.cos(9 * s), np.sin(5 * s)], 1).astype(F32)[None]
    out, counts = nfopp.PathPostprocessor(distance_step=0.5).process_batch(long)
    want = orc.path_postprocess(long[0], 0.001, 0.5)
    assert counts[0] == len(want) and max_abs(out[0, :len(want)].cpu().numpy(), want) < 1e-10
    with pytest.raises(nfopp.NfoppError):
        nfopp.PathPostprocessor().process_batch(np.zeros((1, 1027, 3), F32))
