"""CPU checks of the counter-based sampling restatement (oracle side of csrc/sampling.hip): the exponential-race
resampling must have the distribution of numpy's sequential weighted draw without replacement that the reference uses
(np.random.choice(p=w, replace=False), nfop/nerf_opt_planner.py:131)."""
import numpy as np

from oracle import nfopp_oracle as orc


def test_draws_have_the_right_moments_and_are_counter_addressed():
    u = orc.draw_uniform(11, 5, np.arange(200000), 3, orc.STREAM_T)
    assert 0 <= u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 3e-3
    n = orc.draw_normal(11, 5, np.arange(200000), 3, orc.STREAM_FINE)
    assert abs(n.mean()) < 8e-3 and abs(n.std() - 1) < 8e-3 and np.isfinite(n).all()
    assert abs(np.mean(n ** 4) - 3) < 0.1
    # independent of how many values are requested, and distinct across trajectory / offset / stream
    assert np.array_equal(u[:50], orc.draw_uniform(11, 5, np.arange(50), 3, orc.STREAM_T))
    for other in (orc.draw_uniform(11, 6, np.arange(50), 3, orc.STREAM_T), orc.draw_uniform(11, 5, np.arange(50), 4, orc.STREAM_T),
                  orc.draw_uniform(11, 5, np.arange(50), 3, orc.STREAM_KEY), orc.draw_uniform(12, 5, np.arange(50), 3, orc.STREAM_T)):
        assert not np.array_equal(u[:50], other)


def test_exponential_race_matches_numpy_weighted_choice_without_replacement():
    rng = np.random.default_rng(0)
    C, cap, trials = 12, 5, 4000
    logits = rng.normal(0, 1.5, C).astype(np.float32)
    age = rng.integers(0, 30, C).astype(np.float32)
    w = orc.sigmoid(logits) * np.exp(-0.03 * age) + 1e-6
    w = w / w.sum()
    cand = rng.normal(size=(1, C, 3)).astype(np.float32)
    incl_race = np.zeros(C)
    for trial in range(trials):
        _, _, chosen = orc.resample_pool(cand, age[None], logits[None], cap, seed=99, offset=trial)
        incl_race[chosen[0]] += 1
        assert len(set(chosen[0])) == cap
    np.random.seed(1)
    incl_np = np.zeros(C)
    for _ in range(trials):
        incl_np[np.random.choice(C, cap, replace=False, p=w)] += 1
    # first-order inclusion frequencies agree within sampling noise (3.5 sigma of a binomial)
    p = incl_np / trials
    sigma = np.sqrt(p * (1 - p) / trials * 2) + 1e-3
    assert np.all(np.abs(incl_race / trials - p) < 3.5 * sigma)
    pool, new_age, chosen = orc.resample_pool(cand, age[None], logits[None], cap, seed=1, offset=0)
    assert np.array_equal(pool[0], cand[0, chosen[0]]) and np.array_equal(new_age[0], age[chosen[0]] + 1)


def test_sample_candidates_layout_and_statistics():
    rng = np.random.default_rng(2)
    B, N, D, cap, nf = 3, 40, 3, 30, 10
    prev = rng.uniform(0, 3, (B, N, D)).astype(np.float32)
    bounds = (-0.1, 3.1, -0.2, 3.2)
    cand, age, smp = orc.sample_candidates(prev, None, None, cap, nf, 1.5, 0.02, 0.3, bounds, seed=7, offset=0)
    assert cand.shape == (B, N - 1, D) and smp.shape == (B, N - 1 + cap + nf, D) and (age == 0).all()
    # fine samples hug the segments, course samples scatter with sigma 1.5
    seg_lo = np.minimum(prev[:, 1:], prev[:, :-1]) - 0.1
    seg_hi = np.maximum(prev[:, 1:], prev[:, :-1]) + 0.1
    assert ((cand[..., :2] > seg_lo[..., :2]) & (cand[..., :2] < seg_hi[..., :2])).mean() > 0.97
    assert 1.0 < (smp[:, :N - 1, :2] - cand[..., :2]).std() < 2.0
    field = smp[:, N - 1 + cap:]
    assert (field[..., 0] >= bounds[0]).all() and (field[..., 0] <= bounds[1]).all()
    assert (field[..., 2] >= 0).all() and (field[..., 2] < 2 * np.pi + 1e-6).all()
    # second step: the pool is carried in front of the new fine samples
    pool, page, _ = orc.resample_pool(cand, age, rng.normal(size=(B, N - 1)).astype(np.float32), cap, 7, 0)
    cand2, age2, _ = orc.sample_candidates(prev, pool, page, cap, nf, 1.5, 0.02, 0.3, bounds, seed=7, offset=1)
    assert cand2.shape == (B, cap + N - 1, D)
    assert np.array_equal(cand2[:, :cap], pool) and (age2[:, :cap] == 1).all() and (age2[:, cap:] == 0).all()


def test_grid_checker_restatement():
    grid = np.zeros((30, 40), np.uint8)
    grid[10:20, 5:15] = 255
    xy = np.array([[0.76, 1.26], [0.2, 0.2], [-5.0, 0.0], [3.96, 1.0], [1.0, 2.96]], np.float32)
    got = orc.grid_check(xy, grid, origin_x=0.0, origin_y=0.0, cell=0.1)
    assert got.tolist() == [True, False, True, True, True]
