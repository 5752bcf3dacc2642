"""CPU: the arithmetic behind the bf16x3 split matrix path (csrc/onf_split.hip, csrc/onf_layout.h `split_level`),
emulated in numpy: the three-level split is EXACT, and six partial products are fp32-faithful."""
import numpy as np

F32 = np.float32


def split3(x):
    """x = hi + mid + lo, every level the top 16 bits (a bf16) of the running residual -- `split_level` x3."""
    x = np.asarray(x, F32)

    def top(v):
        return (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(F32)
    hi = top(x)
    r = (x - hi).astype(F32)
    mid = top(r)
    lo = (r - mid).astype(F32)
    return hi, mid, lo


def test_split_is_exact_and_every_level_is_a_bf16():
    rng = np.random.default_rng(0)
    # (residuals of values below ~1e-33 become subnormal and lose their last bits: irrelevant for network weights)
    x = np.concatenate([rng.normal(0, 1, 200000), rng.normal(0, 1e-20, 1000), rng.normal(0, 1e20, 1000),
                        [0.0, -0.0, 1.0, -1.0, 3.4e38, 1e-30, np.pi]]).astype(F32)
    x = x[(np.abs(x) > 1e-30) | (x == 0)]
    hi, mid, lo = split3(x)
    for level in (hi, mid, lo):
        assert np.all((level.view(np.uint32) & np.uint32(0xFFFF)) == 0)        # representable in bf16
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))
    # the (hi | mid) LDS word + the blob's third level rebuild the weight exactly in fp32 (rebuild_weight)
    assert np.array_equal(((hi + mid).astype(F32) + lo).astype(F32), x)


def test_six_partial_products_are_fp32_faithful():
    rng = np.random.default_rng(1)
    n, k = 4000, 224
    w = rng.normal(0, 0.3, (n, k)).astype(F32)
    x = np.sin(rng.normal(0, 3, (n, k))).astype(F32)
    ref = (w.astype(np.float64) * x.astype(np.float64)).sum(1)
    scale = np.abs(w.astype(np.float64) * x).sum(1)
    a, b = split3(w), split3(x)
    kept = [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]
    six = sum((a[i].astype(np.float64) * b[j].astype(np.float64)).sum(1) for i, j in kept)
    seq = np.zeros(n, F32)
    for c in range(k):
        seq = (seq + w[:, c] * x[:, c]).astype(F32)                           # a sequential fp32 dot product
    err_six, err_seq = np.abs(six - ref) / scale, np.abs(seq - ref) / scale
    assert err_six.max() < 2.0 ** -24                                           # dropped terms: below one fp32 rounding
    assert err_six.max() < 0.25 * err_seq.max()
    three = sum((a[i].astype(np.float64) * b[j].astype(np.float64)).sum(1) for i, j in kept[:3])
    assert (np.abs(three - ref) / scale).max() > 1e-6                           # ... and all six are needed
