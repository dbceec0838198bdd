"""Non-statistical float parity checks (BASELINE north star: indices bit-exact on tie-free integer
inputs, float distances within 1e-4 relative).

Two tools:
  * exact brute force: `assert_topk_parity` -- every returned distance is within the tolerance of
    the float64 truth, every returned point is a top-K point up to the tolerance, and wherever the
    id differs from the oracle's the two candidates are a NEAR-TIE in float64 (that is the only
    way two correct exact searches can disagree);
  * traversals (query / merge / top / sym): the oracle is run with the product kernels' float
    SUMMATION ORDER (`orc.wave_order()`), under which ids, distances and counters must agree bit
    for bit -- a decision that depends on a rounding is then the same decision on both sides, and
    nothing statistical is left.  What the different order costs against the reference's own
    (restated, unpinned) cub::BlockReduce order is bounded separately by `assert_order_tolerance`
    (CPU, tests/test_oracle.py)."""
import numpy as np

RTOL = 1e-4        # BASELINE.json north_star: "within 1e-4 relative on float distances"
U = 2.0 ** -24


def cos_atol(D):
    """absolute term for |1 - cos|: the value is a difference of two numbers near 1, so its
    rounding error is absolute -- about sqrt(D) accumulated roundings of relative size 2^-24 on the
    normalised dot product (random-walk growth; the worst case D * 2^-24 is never approached)"""
    return 8.0 * U * np.sqrt(D)


def true_distances(base, q, measure):
    b = base.astype(np.float64)
    x = q.astype(np.float64)
    if measure == 0:
        return (b * b).sum(1)[None, :] - 2.0 * (x @ b.T) + (x * x).sum(1)[:, None]
    nb = np.sqrt((b * b).sum(1))
    nq = np.sqrt((x * x).sum(1))
    den = nq[:, None] * nb[None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        c = np.where(den > 0, np.abs(1.0 - (x @ b.T) / den), 1.0)
    return c


def assert_topk_parity(base, q, ids, d, o_ids, K, measure, what=""):
    """ids, d: [Nq, K] of the engine; o_ids: the oracle's ids for the same queries"""
    D = base.shape[1]
    atol = cos_atol(D) if measure else 0.0
    t = true_distances(base, q, measure)
    if measure == 0:
        # squared L2 through the expanded form in float64: its own error is ~1e-16 * |x|^2
        atol = 1e-9 * float(np.abs(t).max())
    Nq = q.shape[0]
    rows = np.arange(Nq)[:, None]
    valid = ids >= 0
    t_g = np.where(valid, t[rows, np.where(valid, ids, 0)], np.inf)
    kth = np.sort(t, axis=1)[:, min(K, t.shape[1]) - 1][:, None]
    # (a) every distance within the tolerance of the float64 truth
    fin = valid & np.isfinite(d)
    assert np.all(np.abs(d[fin] - t_g[fin]) <= RTOL * np.abs(t_g[fin]) + atol), what
    # (b) sorted, distinct ids
    assert np.all(np.diff(np.where(valid, d, np.inf), axis=1) >= 0), what
    for r in range(Nq):
        v = ids[r][ids[r] >= 0]
        assert len(set(v.tolist())) == len(v), (what, r)
    # (c) every returned point is a top-K point up to the tolerance
    bound = np.broadcast_to(kth * (1 + RTOL) + atol, t_g.shape)
    assert np.all(t_g[valid] <= bound[valid]), what
    # (d) where the id differs from the oracle's, the two candidates are a near-tie in float64
    diff = valid & (o_ids >= 0) & (ids != o_ids)
    if diff.any():
        t_o = t[rows, np.where(o_ids >= 0, o_ids, 0)]
        gap = np.abs(t_g - t_o)[diff]
        scale = np.maximum(np.abs(t_g), np.abs(t_o))[diff]
        assert np.all(gap <= 2 * RTOL * scale + 2 * atol), (what, float(gap.max()))
    return int(diff.sum())


def assert_order_tolerance(d_wave, d_ref, same, D, measure, what=""):
    """distances of identical ids under the two summation orders agree to the contract"""
    atol = cos_atol(D) if measure else 0.0
    a, b = d_wave[same], d_ref[same]
    fin = np.isfinite(a) & np.isfinite(b)
    assert np.all(np.abs(a[fin] - b[fin]) <= RTOL * np.abs(b[fin]) + atol), what
