"""CPU check of the rounding bound behind the exactness certificate of the MFMA bf_query path
(ggnn_amd/csrc/bf_mfma.hip, "Why the pre-selection cannot lose a neighbour"): the expanded form
|a|^2 + |b|^2 - 2 a.b evaluated in float32 on centred rows stays within
E = 1.01 (2D+8) u (|a|^2+|b|^2) of the true squared distance of the ORIGINAL rows, and the direct
form is never below true * (1 - 1.01 (D+3) u); cosine: both forms within (2D+8) u of the truth.
float32 evaluation orders tried: numpy pairwise sums, strictly sequential sums, reversed order."""
import numpy as np
import pytest

U = 2.0 ** -24


def _clustered(N, D, seed, offset=0.0, scale=1.0):
    rng = np.random.default_rng(seed)
    centres = rng.normal(size=(32, D)) * 4.0
    x = centres[rng.integers(0, 32, N)] + rng.normal(size=(N, D))
    return (x * scale + offset).astype(np.float32)


def _seq_sum32(v):
    acc = np.float32(0.0)
    for x in v.astype(np.float32):
        acc = np.float32(acc + x)
    return acc


MAKERS = {
    "offset1000": lambda N, D, s: _clustered(N, D, s, 1000.0),
    "tiny": lambda N, D, s: _clustered(N, D, s, -5.0, 1e-3),
    "uni+1000": lambda N, D, s: np.random.default_rng(s).random((N, D), dtype=np.float32) + np.float32(1000),
    "wide": lambda N, D, s: (_clustered(N, D, s) * 10.0 ** np.random.default_rng(s + 1).uniform(-3, 3, (N, 1))).astype(np.float32),
    "ints": lambda N, D, s: np.random.default_rng(s).integers(0, 256, (N, D)).astype(np.float32),
}


@pytest.mark.parametrize("name", list(MAKERS))
@pytest.mark.parametrize("D", [32, 128, 960])
@pytest.mark.parametrize("center", [True, False])
def test_expanded_form_error_bound_l2(name, D, center):
    N, Nq = 300, 12
    base, q = MAKERS[name](N, D, 7), MAKERS[name](Nq, D, 8)
    mu = base[::3].mean(0).astype(np.float32) if center else np.zeros(D, np.float32)
    a, b = (q - mu).astype(np.float32), (base - mu).astype(np.float32)   # fl(x - mu), elementwise
    true = ((base.astype(np.float64)[None] - q.astype(np.float64)[:, None]) ** 2).sum(2)
    an64, bn64 = (a.astype(np.float64) ** 2).sum(1), (b.astype(np.float64) ** 2).sum(1)
    for order in ("pairwise", "sequential", "reversed"):
        for i in range(Nq):
            for j in range(0, N, 17):
                pa, pb, pab = a[i] * a[i], b[j] * b[j], a[i] * b[j]          # float32 products
                if order == "pairwise":
                    qn, bn, dot = pa.sum(dtype=np.float32), pb.sum(dtype=np.float32), pab.sum(dtype=np.float32)
                elif order == "sequential":
                    qn, bn, dot = _seq_sum32(pa), _seq_sum32(pb), _seq_sum32(pab)
                else:
                    qn, bn, dot = _seq_sum32(pa[::-1]), _seq_sum32(pb[::-1]), _seq_sum32(pab[::-1])
                d_e = np.float32(np.float32(qn + bn) - np.float32(2) * dot)
                E = 1.01 * (2 * D + 8) * U * (float(qn) + float(bn))
                assert abs(float(d_e) - true[i, j]) <= E, (order, i, j, float(d_e), true[i, j], E)
                # computed norms are close enough to the exact ones for the bound's 1.01 factor
                assert abs(float(qn) - an64[i]) <= 2 * D * U * an64[i] + 1e-300
                assert abs(float(bn) - bn64[j]) <= 2 * D * U * bn64[j] + 1e-300
                diff = (base[j] - q[i]).astype(np.float32)
                d_dir = _seq_sum32(diff * diff)
                assert float(d_dir) >= true[i, j] * (1 - 1.01 * (D + 3) * U)


@pytest.mark.parametrize("name", ["wide", "offset1000", "ints"])
@pytest.mark.parametrize("D", [128, 960])
def test_cosine_forms_within_absolute_bound(name, D):
    N, Nq = 300, 12
    base, q = MAKERS[name](N, D, 17), MAKERS[name](Nq, D, 18)
    b64, q64 = base.astype(np.float64), q.astype(np.float64)
    for i in range(Nq):
        for j in range(0, N, 17):
            true = abs(1.0 - (b64[j] @ q64[i]) / (np.linalg.norm(b64[j]) * np.linalg.norm(q64[i])))
            for rev in (False, True):
                sl = slice(None, None, -1) if rev else slice(None)
                dot = _seq_sum32((base[j] * q[i])[sl])
                qn, bn = _seq_sum32((q[i] * q[i])[sl]), _seq_sum32((base[j] * base[j])[sl])
                d = np.abs(np.float32(1) - np.float32(dot / np.sqrt(np.float32(qn * bn))))
                assert abs(float(d) - true) <= (2 * D + 8) * U
