"""CPU: the gate of bench.py's parity block (thresholds from tests/golden/g17_conditioning.npz) -- passes at the levels MI355X
measures, fails on a wrong-gradient-sized deviation, on a broken first step and on a blown tail."""
import numpy as np

import bench


def _diff(rng, median=1e-5, frac_big=0.01, big=0.03):
    d = np.abs(rng.normal(0, median * 1.4826, (256, 256, 3)))
    mask = rng.uniform(size=d.shape) < frac_big
    d[mask] = rng.uniform(0, big, mask.sum())
    return d


def test_gate_passes_at_measured_levels_and_fails_on_regressions():
    rng = np.random.default_rng(0)
    terms = {k: 1e-7 for k in ("total", "distance", "softplus_sum")}
    ok = bench.parity_gate(_diff(rng), terms, 10)
    assert ok["ok"] and ok["checks"]["xy"]["ok"] and ok["checks"]["theta"]["ok"]
    # a wrong tap / stale weight image moves the bulk: median 1e-3
    assert not bench.parity_gate(_diff(rng, median=1e-3), terms, 10)["ok"]
    # a wrong first step shows in the per-term sums
    assert not bench.parity_gate(_diff(rng), dict(terms, softplus_sum=3e-5), 10)["ok"]
    # the tail: one entry far beyond the reference's own spread
    d = _diff(rng)
    d[7, 100, 0] = 2.0
    assert not bench.parity_gate(d, terms, 10)["ok"]
    # p99 beyond 4 x the reference's 90th percentile
    assert not bench.parity_gate(_diff(rng, frac_big=0.05, big=0.5), terms, 10)["ok"]
    # beyond the conditioning file's horizon only the first-step terms gate
    long = bench.parity_gate(_diff(rng, median=1e-3), terms, 50)
    assert long["ok"] and set(long["checks"]) == {"first_step_terms_rel"}
