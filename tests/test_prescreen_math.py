"""CPU check of the arithmetic behind the exact pre-screen (ggnn_amd/csrc/traversal.hpp,
prescreen.hip): a float32 restatement of the coding and of the threshold formula, checked
against float32 distances on adversarial value ranges.  The GPU tests check the kernels
themselves (ggnn_op_prescreen_probe); this one checks that the margins in the derivation hold
in float32 arithmetic at all."""
import numpy as np
import pytest

F = np.float32
U = F(2.0 ** -24)


def encode(base, power_of_two):
    """ps_minmax/finalize/encode/retry kernels"""
    o = base.min(0).astype(F)
    rng_ = F((base.max(0) - o).max())
    if power_of_two:
        s = F(2.0) ** np.ceil(np.log2(rng_ / F(255))) if rng_ > 0 else F(1)
        while F(255) * s < rng_:
            s = s * F(2)
    else:
        s = rng_ / F(255) * (F(1) + F(2.0 ** -20))
    s = F(s)
    inv_s = F(1) / s
    c = np.clip(np.rint((base - o) * inv_s), 0, 255).astype(F)
    res = base.astype(np.float64) - (o.astype(np.float64) + np.float64(s) * c)
    e_rows = np.sqrt((res ** 2).sum(1))
    o_norm = F(np.sqrt((o.astype(F) ** 2).sum(dtype=F))) * (F(1) + F(1e-6))
    e_max = F(0)
    if e_rows.max() > 0:
        e_max = F(e_rows.max()) * (F(1) + F(1e-6)) + F(1e-12) * (o_norm + F(255) * s *
                                                                 F(np.sqrt(base.shape[1])))
    return c.astype(np.int64), o, s, inv_s, F(e_max), o_norm


def query_side(q, o, s, inv_s, e_max, o_norm, Dc):
    """Prescreen::load"""
    t = ((q - o) * inv_s).astype(F)
    code = np.clip(np.rint(t), 0, 255).astype(F)
    diff = (t - code).astype(F)
    eq = F((diff * diff).sum(dtype=F))
    m = F(4) * F(Dc + 32) * U
    q_norm = F(np.sqrt((q * q).sum(dtype=F)))
    e_q = s * F(np.sqrt(eq)) * (F(1) + m)
    slack = e_q + F(8) * U * (q_norm + o_norm) + e_max
    return code.astype(np.int64), F(slack), m


def threshold(crit, slack, inv_s, m):
    """Prescreen::threshold (squared L2); 1-ulp square root like v_sqrt_f32"""
    t = F(np.sqrt(crit)) * (F(1) - F(2) * U)  # pessimistic: a result one ulp too small
    t = t * (F(1) + m) + slack
    t = t * inv_s * (F(1) + m)
    return t * t * (F(1) + m)


def float32_distance_lower(q, x):
    """smallest value a float32 evaluation of sum (q-x)^2 can return (any summation order)"""
    d = ((q.astype(np.float64) - x.astype(np.float64)) ** 2).sum()
    return d * (1.0 - (q.size + 8) * 2.0 ** -24)


CASES = {
    "integers": lambda r, n, d: r.integers(0, 256, (n, d)).astype(F),
    "fractional": lambda r, n, d: (r.normal(size=(n, d)) * 37.5 + 128).astype(F),
    "large offset": lambda r, n, d: (r.normal(size=(n, d)) + 1.0e5).astype(F),
    "tiny scale": lambda r, n, d: (r.normal(size=(n, d)) * 1e-4 - 3).astype(F),
    "one wide dimension": lambda r, n, d: np.concatenate(
        [r.normal(size=(n, 1)) * 1e3, r.normal(size=(n, d - 1))], 1).astype(F),
    "negative": lambda r, n, d: (-np.abs(r.normal(size=(n, d))) * 50).astype(F),
}


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("power_of_two", [True, False])
def test_a_rejected_candidate_is_at_least_as_far_as_the_criteria(name, power_of_two):
    rng = np.random.default_rng(hash(name) % 1000)
    N, D, Nq = 400, 64, 40
    base = CASES[name](rng, N, D)
    queries = CASES[name](rng, Nq, D)
    queries[0] = base[0]                      # a query equal to a base row
    queries[1] = queries[1] * F(1.5) + F(7)   # a query outside the coded range
    codes, o, s, inv_s, e_max, o_norm = encode(base, power_of_two)
    checked = rejected = 0
    for q in queries:
        cq, slack, m = query_side(q, o, s, inv_s, e_max, o_norm, D)
        S = ((cq[None, :] - codes) ** 2).sum(1).astype(F)
        for i in range(N):
            d_low = float32_distance_lower(q, base[i])
            # criteria just above the smallest possible float32 evaluation, and a few below it
            for crit in (np.nextafter(F(d_low), F(np.inf)), F(d_low * 0.9), F(d_low * 0.5)):
                if not np.isfinite(crit) or crit < 0:
                    continue
                thr = threshold(crit, slack, inv_s, m)
                checked += 1
                if S[i] >= thr:
                    rejected += 1
                    # rejection is only allowed when every float32 evaluation is >= crit
                    assert d_low >= float(crit), (name, i, float(S[i]), float(thr))
    assert checked > 0
    if name in ("integers", "fractional", "negative"):
        assert rejected > 0.3 * checked / 3  # and the bound is useful, not vacuous


def cosine_distance_lower(q, x):
    """smallest value a float32 evaluation of |1 - q.x / sqrt(|q|^2 |x|^2)| can return"""
    q64, x64 = q.astype(np.float64), x.astype(np.float64)
    nn = (q64 * q64).sum() * (x64 * x64).sum()
    if nn <= 0:
        return 1.0
    d = abs(1.0 - (q64 * x64).sum() / np.sqrt(nn))
    return d - (q.size + 8) * 2.0 ** -24  # absolute: the cancellation in 1 - cos


@pytest.mark.parametrize("name", ["fractional", "large offset", "negative", "integers"])
def test_cosine_variant(name):
    rng = np.random.default_rng(hash(name) % 997)
    N, D, Nq = 300, 64, 30
    base = CASES[name](rng, N, D)
    base[3] = 0  # zero rows stay zero after "normalisation" and have distance 1 to everything
    queries = CASES[name](rng, Nq, D)
    norms = np.sqrt((base.astype(np.float64) ** 2).sum(1))
    unit = np.where(norms[:, None] > 0, base / np.maximum(norms, 1e-300)[:, None], 0.0)
    # codes are measured against the exactly normalised rows (prescreen.hip, cosine mode)
    codes, o, s, inv_s, _, o_norm = encode(unit.astype(F), False)
    res = unit - (o.astype(np.float64) + np.float64(s) * codes)
    e_max = F(np.sqrt((res ** 2).sum(1)).max()) * (F(1) + F(1e-6)) + F(1e-12) * (
        o_norm + F(255) * s * F(np.sqrt(D)))
    rejected = 0
    for q in queries:
        qn = F(np.sqrt((q * q).sum(dtype=F)))
        qu = (q * (F(1) / qn)).astype(F)
        cq, slack, m = query_side(qu, o, s, inv_s, e_max, o_norm, D)
        # Prescreen::load uses |q^| = 1 for the slack of the cosine mode
        slack = slack - F(8) * U * F(np.sqrt((qu * qu).sum(dtype=F))) + F(8) * U
        S = ((cq[None, :] - codes) ** 2).sum(1).astype(F)
        for i in range(N):
            d_low = cosine_distance_lower(q, base[i])
            for crit in (F(max(d_low, 0.0)) + F(1e-7), F(max(d_low, 0.0) * 0.5)):
                c2 = F(2) * (crit + m) * (F(1) + m)
                if S[i] >= threshold(c2, slack, inv_s, m):
                    rejected += 1
                    assert d_low >= float(crit), (name, i)
    if name != "large offset":  # there all vectors are parallel: distances below the margin
        assert rejected > 0
