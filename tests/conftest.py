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


# Gates for the fixtures made with the benchmarked settings (tests/golden/traj_benchmr_*.npz, g14: scripts/run_bench_mr.py
# hyper block -- w_col 100, beta 10, w_dir 100, lr 5e-2 -- on the 100 m random-disc map).  That configuration is far
# stiffer than scripts/benchmark.py's, so the gates are DERIVED from the reference's own conditioning, measured by
# tools/ref_conditioning.py (build container: the reference restarted from the fixture state with every trajectory
# coordinate moved by ONE fp32 ulp, 6 random sign patterns, distance to the unperturbed run) and stored in
# tests/golden/g17_conditioning.npz:
#   rollouts  gate = 4 x that spread per quantity (xy, theta, lambda, cm) up to 10 steps, 8 x after 50 (a different
#             summation order perturbs every step by an ulp, not only the start)
#   g14       (B = 4 from the straight-line start: zero-up-to-rounding gradient entries become +-lr Adam moves, so single
#             entries flip in the reference itself -- 2 lr = 0.1 after ONE step) maximum gate = 1.5 x the spread's
#             maximum; the bulk (99th / 90th percentile) is a regression gate at about 5 x what tools/gpu_benchmr_margins.py
#             measures on MI355X (its output: profiles/r03_benchmr_margins.txt) -- far below the reference's own bulk spread
CONDITIONING = load_golden("g17_conditioning.npz")
BENCHMR_FIXTURES = [("traj_benchmr_n256.npz", (1, 10, 50)), ("traj_benchmr_n512.npz", (1, 10))]


def benchmr_rollout_tol(name, K):
    tag = name.replace("traj_benchmr_", "").replace(".npz", "")
    s = CONDITIONING["rollout_%s_k%d" % (tag, K)] * (4.0 if K <= 10 else 8.0)
    return dict(xy=float(s[0]), th=float(s[1]), lam=float(s[2]), cm=float(s[3]))


# Bulk regression gate of the rollouts (the conditioning-derived maxima above leave 25 x headroom on xy at 50
# steps -- a wrong tap or a stale weight image for a few steps would pass): 90th percentile of |difference| per quantity,
# about 3 x what MI355X measures against these fixtures (tools/gpu_benchmr_margins.py -> profiles/r03_benchmr_margins.txt:
# n256 k50 p90 = 4.1e-3 / 2.4e-3 / 1.3e-3 / 1.7e-6, k10 = 7.6e-6 / 4.4e-6 / 2.7e-6 / 3e-9, n512 k10 = 1.1e-5 / 2.6e-6 / 2.7e-6 / 0;
# k1 = one ulp of a coordinate near 96 m).  The maxima stay the hard limit.
_ROLLOUT_BULK_P90 = {1: dict(xy=2.5e-5, th=1e-5, lam=5e-6, cm=1e-7), 10: dict(xy=5e-5, th=2.5e-5, lam=1.5e-5, cm=1e-7),
                     50: dict(xy=1.3e-2, th=7.5e-3, lam=4e-3, cm=1e-5)}


def check_benchmr_rollout(name, K, traj, lam, cm, z, bulk=True):
    """maximum gate (conditioning-derived) + p90 bulk gate of one rollout snapshot `g6_k<K>_*` of a bench-mr fixture"""
    pre, tol = "g6_k%d_" % K, benchmr_rollout_tol(name, K)
    for key, got, ref in (("xy", traj[:, :2], z[pre + "traj"][:, :2]), ("th", traj[:, 2], z[pre + "traj"][:, 2]),
                          ("lam", lam, z[pre + "lam"]), ("cm", cm, z[pre + "cm"])):
        assert max_abs(got, ref) < tol[key], (name, K, key, max_abs(got, ref))
        if bulk:
            assert abs_percentile(got, ref, 90) <= _ROLLOUT_BULK_P90[K][key], (name, K, key, abs_percentile(got, ref, 90))


_BATCH_BULK = {1: dict(q=99, xy=1e-5, th=2e-5, lam=1e-7, cm=1e-7), 3: dict(q=99, xy=5e-3, th=5e-3, lam=2e-4, cm=1e-7),
               12: dict(q=90, xy=2e-3, th=1e-3, lam=4e-4, cm=1e-5)}
BENCHMR_BATCH_TOL = {}
for _k, _bulk in _BATCH_BULK.items():
    _mx = CONDITIONING["batch_k%d_max" % _k] * 1.5
    assert int(CONDITIONING["batch_k%d_q" % _k]) == _bulk["q"]
    BENCHMR_BATCH_TOL[_k] = dict(q=_bulk["q"], **{name: (_bulk[name], float(_mx[i])) for i, name in enumerate(("xy", "th", "lam", "cm"))})


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
