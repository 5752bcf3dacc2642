import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "pytorch-motion-planner_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def max_rel(a, b, floor=1e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / (np.abs(b) + floor)))


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


# Gates for the fixtures made with the benchmarked settings (tests/golden/traj_benchmr_*.npz: scripts/run_bench_mr.py
# hyper block -- w_col 100, beta 10, w_dir 100, lr 5e-2 -- on the 100 m random-disc map).  That configuration is far
# stiffer than scripts/benchmark.py's: the reference itself, restarted from the fixture's state with every coordinate
# moved by ONE fp32 ulp, ends (xy, theta, lambda, cm) = (1.5e-5, 3e-6, 2e-6, 1e-8) away after 1 step at N=256
# [(3.8e-5, 1.0e-4, 2.7e-5, 7e-7) at N=512], (1.5e-3, 5e-5, 5e-5, 9e-7) after 10 and (8.4e-3, 6.7e-3, 2.7e-3, 6e-5)
# after 50 steps (measured in the build container, torch 2.10 CPU).  Gates = about 4x that conditioning; cm after 50
# steps 10x (one perturbed ulp is the FLOOR of what a different summation order does in every step: the fp32 matrix path
# measured 3.5e-4 there, the split path 2.9e-5, both with collision samples that are the reference's bit for bit;
# tools/gpu_benchmr_margins.py prints all of these).
BENCHMR_ROLLOUT_TOL = {1: dict(xy=1.5e-4, th=4e-4, lam=1e-4, cm=3e-6),
                       10: dict(xy=6e-3, th=2e-3, lam=1e-3, cm=1e-5),
                       50: dict(xy=3e-2, th=3e-2, lam=1e-2, cm=6e-4)}
BENCHMR_FIXTURES = [("traj_benchmr_n256.npz", (1, 10, 50)), ("traj_benchmr_n512.npz", (1, 10))]
# g14 (B = 4, 12 steps FROM the straight-line initialisation, snapshots after steps 1 / 3 / 12).  On a straight line many
# gradient entries are zero up to rounding and Adam's first steps turn each into a full +-lr move, so single entries are
# ill-conditioned in the reference itself: restarted 1 ulp away it ends max (xy, theta, lambda, cm) = (0.10, 0.10, 2e-6,
# 2e-8) away after ONE step (= 2 lr: a sign flip), (0.09, 0.14, 0.026, 4e-8) after 3 and (0.14, 0.23, 0.035, 8.6e-4)
# after 12.  Gates: the bulk of the entries (99th / 90th percentile of |difference|) tightly, the maximum at that
# conditioning.  Measured on MI355X (tools/gpu_benchmr_margins.py): p99 0 / 9e-4 (xy, steps 1 / 3), p90 4e-4 (step 12).
BENCHMR_BATCH_TOL = {
    1: dict(q=99, xy=(1e-5, 0.11), th=(2e-5, 0.11), lam=(1e-7, 1e-5), cm=(1e-7, 1e-6)),
    3: dict(q=99, xy=(5e-3, 0.15), th=(5e-3, 0.2), lam=(2e-4, 0.05), cm=(1e-7, 1e-6)),
    12: dict(q=90, xy=(2e-3, 0.15), th=(1e-3, 0.25), lam=(4e-4, 0.05), cm=(1e-5, 1e-3)),
}


def abs_percentile(a, b, q):
    return float(np.percentile(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)), q))


def check_batch_snapshot(k, traj, lam, cm, z):
    """(percentile gate, maximum gate) per quantity of BENCHMR_BATCH_TOL[k] against snapshot k of g14."""
    tol = BENCHMR_BATCH_TOL[k]
    for name, got, ref in (("xy", traj[..., :2], z["k%d_traj" % k][..., :2]), ("th", traj[..., 2], z["k%d_traj" % k][..., 2]),
                           ("lam", lam, z["k%d_lam" % k]), ("cm", cm, z["k%d_cm" % k])):
        bulk, worst = tol[name]
        assert abs_percentile(got, ref, tol["q"]) <= bulk, (k, name, abs_percentile(got, ref, tol["q"]))
        assert max_abs(got, ref) <= worst, (k, name, max_abs(got, ref))


@pytest.fixture(autouse=True)
def _matrix_path_is_left_as_found():
    """The matrix-path switch is process-wide state of libnfopp_hip.so: whatever a test does, the next test starts on
    the path this one started on (a module once left the process on the non-default kernel)."""
    try:
        from nfopp import _lib
        lib = _lib.load()
    except Exception:
        yield
        return
    before = lib.nfopp_get_matrix_path()
    yield
    if lib.nfopp_get_matrix_path() != before:
        lib.nfopp_set_matrix_path(before)
