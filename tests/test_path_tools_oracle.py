"""CPU: oracle restatements of the steps either side of the planner step against outputs
of the reference itself (tests/golden/g12, made by make_golden.py with scipy 1.15 / numpy 2.2 / torch 2.10)."""
import numpy as np
import pytest

from conftest import load_golden, max_abs
from oracle import nfopp_oracle as orc


@pytest.mark.parametrize("n", [50, 51])
def test_directed_initialiser_vs_reference(n):
    z = load_golden("g12_init_dir_postprocess.npz")
    for c, want in zip(z["dir_cases"], z["dir_traj_n%d" % n]):
        got = orc.initialize_trajectory_directed(c[:3], c[3:], n)
        assert np.array_equal(got[:, :2], want[:, :2])          # xy: torch.linspace reproduced bit for bit
        assert max_abs(got[:, 2], want[:, 2]) < 1e-6            # headings: atan2 rounding (numpy vs torch)


def test_path_postprocess_vs_reference():
    z = load_golden("g12_init_dir_postprocess.npz")
    for i in range(5):
        got = orc.path_postprocess(z["post_in_%d" % i])
        want = z["post_out_%d" % i]
        assert got.shape == want.shape and got.dtype == np.float64
        assert max_abs(got, want) < 1e-12                       # float64 spline; scipy solves with LAPACK gbsv
    got = orc.path_postprocess(z["post_in_1"], *z["post_alt_params"])
    assert got.shape == z["post_out_alt_1"].shape and max_abs(got, z["post_out_alt_1"]) < 1e-12
    # the reversing start of path 1 is trimmed: more than the default single pose is dropped
    full = orc.path_postprocess(z["post_in_0"])
    assert len(full) == len(z["post_out_0"])
    short = z["post_in_0"][:2]
    assert np.array_equal(orc.path_postprocess(short), short)   # < 3 poses pass through (:14-15)
    with pytest.raises(ValueError):
        orc.path_postprocess(np.zeros((5, 3), np.float32))      # all poses coincide -> 2 survive -> no spline


def test_pairwise_sum_matches_numpy():
    rng = np.random.default_rng(0)
    for n in (1, 7, 8, 9, 127, 128, 129, 255, 300, 1025):
        a = rng.uniform(0, 1, n).astype(np.float32)
        assert orc._pairwise_sum_f32(a) == np.sum(a)
