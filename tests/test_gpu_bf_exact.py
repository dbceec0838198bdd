"""bf_query on the matrix cores must be EXACT -- it is a hot-path function and the recall ground
truth (reference: BruteForceQueryKernel, src/ggnn/query/bf_query_layer.cu:52-57, exact by
construction).  The MFMA path pre-selects with the expanded distance form, certifies every query
with a rounding bound and re-scans the ones it cannot certify (ggnn_amd/csrc/bf_mfma.hip).

Checked here on data chosen to break an un-certified pre-selection (large common offsets, tiny
scales, rows of very different norms, tight far-away clusters, an outlier row):
  * bit-identical ids and distances to the scan kernel (same direct-form arithmetic, taken for
    batches of < 256 queries),
  * agreement with the CPU oracle up to float rounding of near-ties, judged in float64,
  * the certificate / re-scan counters behave as designed (centred data is certified, the
    un-centred test hook and degenerate data are re-scanned -- and still exact).
"""
import os

import numpy as np
import pytest

from conftest import make_int_data, make_uni_data

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from ggnn_amd import ops as o
    return o


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _clustered(N, D, seed, offset=0.0, scale=1.0):
    rng = np.random.default_rng(seed)
    centres = rng.normal(size=(32, D)) * 4.0
    x = centres[rng.integers(0, 32, N)] + rng.normal(size=(N, D))
    return (x * scale + offset).astype(np.float32)


def _uni_offset(N, D, seed):
    return make_uni_data(N, D, seed) + np.float32(1000.0)


def _mixed_norms(N, D, seed):
    """rows of very different lengths (cosine must not care; L2 norms span 12 decades)"""
    rng = np.random.default_rng(seed)
    x = _clustered(N, D, seed)
    return (x * (10.0 ** rng.uniform(-3, 3, (N, 1)))).astype(np.float32)


def _far_tight(N, D, seed):
    """tight clusters (spread 1e-3) around far-apart centres: neighbour distances ~1e-4 next to
    squared norms ~1e3 even after centring -> cannot be certified, must be re-scanned"""
    rng = np.random.default_rng(seed)
    centres = np.random.default_rng(99).normal(size=(8, D)) * 4.0  # the same for base and queries
    x = centres[rng.integers(0, 8, N)] + 1e-3 * rng.normal(size=(N, D))
    return x.astype(np.float32)


def _with_outlier(N, D, seed):
    x = _clustered(N, D, seed)
    if N > 5000:  # the base only: one row with an enormous norm inflates the bound for everyone
        x[1234] = 1.0e6
    return x


def scan_answer(ops, b, q, K, measure):
    """the scan kernel (batches of < 256 queries never take the MFMA path)"""
    ids, dists = [], []
    for i in range(0, q.shape[0], 128):
        a, d = ops.bf_query(b, q[i:i + 128].contiguous(), K, measure)
        ids.append(a)
        dists.append(d)
    return torch.cat(ids), torch.cat(dists)


def check_against_float64(base, q, ids, K, measure, rel=2e-6):
    """no true neighbour may be missing except within float rounding of the K-th distance"""
    b64, q64 = base.astype(np.float64), q.astype(np.float64)
    for n in range(0, q.shape[0], 7):
        if measure == 0:
            d = ((b64 - q64[n]) ** 2).sum(1)
        else:
            nb, nq = np.linalg.norm(b64, axis=1), np.linalg.norm(q64[n])
            d = np.abs(1.0 - (b64 @ q64[n]) / np.maximum(nb * nq, 1e-300))
        got = ids[n]
        assert len(set(got.tolist())) == K
        kth = np.sort(d)[K - 1]
        tol = rel * max(kth, 1e-30) + (1e-6 if measure else 0.0)
        assert d[got].max() <= kth + tol, (n, d[got].max(), kth)
        # and the rows are in ascending order up to the same rounding
        assert (np.diff(d[got]) >= -tol).all()


CASES = [
    ("uni+1000", _uni_offset, 0, 128), ("clustered+1000", lambda N, D, s: _clustered(N, D, s, 1000.0), 0, 128),
    ("clustered*1e-3-5", lambda N, D, s: _clustered(N, D, s, -5.0, 1e-3), 0, 128),
    ("clustered+1000 D=96", lambda N, D, s: _clustered(N, D, s, 1000.0), 0, 96),
    ("clustered+1000 D=200", lambda N, D, s: _clustered(N, D, s, 1000.0), 0, 200),
    ("clustered+300 D=960", lambda N, D, s: _clustered(N, D, s, 300.0), 0, 960),
    ("mixed norms cosine", _mixed_norms, 1, 128), ("mixed norms cosine D=960", _mixed_norms, 1, 960),
    ("clustered+1000 cosine", lambda N, D, s: _clustered(N, D, s, 1000.0), 1, 128),
    ("mixed norms L2", _mixed_norms, 0, 128), ("outlier row", _with_outlier, 0, 128),
]


@pytest.mark.parametrize("name,maker,measure,D", CASES, ids=[c[0] for c in CASES])
def test_bf_mfma_adversarial_equals_scan(ops, orc, name, maker, measure, D):
    N, Nq, K = (20000, 300, 10) if D <= 256 else (8000, 256, 10)
    base, q = maker(N, D, 501), maker(Nq, D, 502)
    b, qq = dev(base), dev(q)
    ids, dists, rescanned = ops.bf_query(b, qq, K, measure, rescanned=True)
    s_ids, s_dists = scan_answer(ops, b, qq, K, measure)
    assert torch.equal(ids, s_ids), name
    assert torch.equal(dists, s_dists), name
    check_against_float64(base, q, ids.cpu().numpy(), K, measure)
    # the oracle agrees wherever its own (differently ordered) float sums do not reorder near-ties
    o_ids, o_d = orc.bf_query(base, q[:64], K, measure)
    np.testing.assert_allclose(dists[:64].cpu().numpy(), o_d, rtol=1e-4, atol=1e-6 if measure else 0)
    if name in ("uni+1000", "clustered+1000", "clustered*1e-3-5", "clustered+1000 D=96"):
        assert rescanned == 0, (name, rescanned)   # centring makes these certifiable


def test_bf_mfma_uncertifiable_data_is_rescanned(ops, orc):
    """tight far clusters: the bound cannot separate the K-th from the (K+8)-th neighbour, so the
    queries are answered by the scan kernel -- correct, just not fast"""
    N, Nq, K, D = 20000, 300, 10, 128
    base, q = _far_tight(N, D, 511), _far_tight(Nq, D, 512)
    b, qq = dev(base), dev(q)
    ids, dists, rescanned = ops.bf_query(b, qq, K, 0, rescanned=True)
    s_ids, s_dists = scan_answer(ops, b, qq, K, 0)
    assert torch.equal(ids, s_ids) and torch.equal(dists, s_dists)
    assert rescanned > Nq // 2
    check_against_float64(base, q, ids.cpu().numpy(), K, 0, rel=1e-5)


def test_bf_mfma_without_centring_falls_back_and_stays_exact(ops):
    """GGNN_BF_NO_CENTER=1 (test hook): offset data is no longer certifiable -> every query is
    re-scanned; with centring none is.  Results identical in all three runs."""
    N, Nq, K, D = 20000, 300, 10, 128
    base, q = _clustered(N, D, 521, 1000.0), _clustered(Nq, D, 522, 1000.0)
    b, qq = dev(base), dev(q)
    ids, dists, r0 = ops.bf_query(b, qq, K, 0, rescanned=True)
    os.environ["GGNN_BF_NO_CENTER"] = "1"
    try:
        ids2, dists2, r1 = ops.bf_query(b, qq, K, 0, rescanned=True)
    finally:
        del os.environ["GGNN_BF_NO_CENTER"]
    s_ids, s_dists = scan_answer(ops, b, qq, K, 0)
    assert r0 == 0 and r1 == Nq
    for a, d in ((ids, dists), (ids2, dists2)):
        assert torch.equal(a, s_ids) and torch.equal(d, s_dists)


@pytest.mark.parametrize("K", [1, 10, 100])
def test_bf_mfma_integer_data_certified_and_exact(ops, orc, K):
    """the S-int track: still bit-identical to the oracle, and nothing needs the re-scan"""
    base, q = make_int_data(30000, 128, 531), make_int_data(300, 128, 532)
    ids, dists, rescanned = ops.bf_query(dev(base), dev(q), K, 0, rescanned=True)
    o_ids, o_d = orc.bf_query(base, q, K)
    assert np.array_equal(ids.cpu().numpy(), o_ids) and np.array_equal(dists.cpu().numpy(), o_d)
    assert rescanned <= 3


def test_bf_mfma_duplicate_rows_ties(ops, orc):
    """every distance occurs three times and some queries are base rows (d = 0): ties at the
    boundary of the candidate lists cannot be certified strictly -> re-scan keeps Q2 order"""
    base = make_int_data(3000, 64, 3)
    base = np.concatenate([base, base, base])
    q = np.concatenate([make_int_data(200, 64, 4), base[:100]])
    ids, dists, rescanned = ops.bf_query(dev(base), dev(q), 12, 0, rescanned=True)
    o_ids, o_d = orc.bf_query(base, q, 12)
    assert np.array_equal(ids.cpu().numpy(), o_ids) and np.array_equal(dists.cpu().numpy(), o_d)


def test_engine_reports_rescanned_queries(orc):
    import ggnn_amd as ggnn
    base, q = _clustered(20000, 128, 541, 1000.0), _clustered(300, 128, 542, 1000.0)
    eng = ggnn.GGNN()
    eng.set_base(base)
    gt, gd = eng.bf_query(q, 10)
    assert eng.last_bf_query_rescanned() == 0
    check_against_float64(base, q, gt.numpy(), 10, 0)


@pytest.mark.parametrize("N,Nq,K,D", [(4500, 257, 1, 128), (4500, 259, 3, 100), (9100, 301, 5, 200)])
def test_bf_mfma_odd_sizes(ops, orc, N, Nq, K, D):
    """odd slice / query / list counts: every scratch block of the MFMA path must stay 16-byte
    aligned (the shifted query copy sits behind the candidate buffers)"""
    base, q = _clustered(N, D, 551, 1000.0), _clustered(Nq, D, 552, 1000.0)
    b, qq = dev(base), dev(q)
    ids, dists, _ = ops.bf_query(b, qq, K, 0, rescanned=True)
    s_ids, s_dists = scan_answer(ops, b, qq, K, 0)
    assert torch.equal(ids, s_ids) and torch.equal(dists, s_dists)
    check_against_float64(base, q, ids.cpu().numpy(), K, 0)


@pytest.mark.parametrize("N,Nq,D,slices", [
    (4100, 110_000, 32, 0),    # 860 query blocks > resident workgroups: ranges span several blocks
    (4100, 40_000, 64, 0),     # a range = a bit more than one query block
    (100_000, 1000, 128, 0),   # ranges floored at 1/32 of a query block
    (30_000, 1500, 96, 7),     # hook: ranges of 1/7 of a query block (not a multiple of the tiles)
    (30_000, 1500, 128, 1),    # hook: one range per query block
])
def test_bf_mfma_equal_tile_ranges(ops, N, Nq, D, slices):
    """the single-chunk kernel cuts the (query block, tile) sequence into equal ranges; a range that
    crosses query-block boundaries is processed in segments with fresh lists, and query blocks have
    different numbers of parts (bf_mfma.hip, "Work of this workgroup"): results must equal the scan
    kernel's, bit for bit, for every way the ranges fall"""
    from ggnn_amd import _lib
    base = make_int_data(N, D, 901)
    q = make_int_data(Nq, D, 902)
    b, qq = dev(base), dev(q)
    with _lib.hooks(**({"BF_SLICES": slices} if slices else {})):
        ids, dists, resc = ops.bf_query(b, qq, 10, 0, rescanned=True)
    # scan kernel in batches (it is the path for < 256 queries)
    sel = np.unique(np.concatenate([np.arange(0, Nq, max(1, Nq // 600)), np.arange(Nq - 130, Nq),
                                    np.arange(120, 140)]))
    sub = dev(q[sel])
    s_ids = torch.cat([ops.bf_query(b, sub[i:i + 200].contiguous(), 10, 0)[0] for i in range(0, len(sel), 200)])
    s_d = torch.cat([ops.bf_query(b, sub[i:i + 200].contiguous(), 10, 0)[1] for i in range(0, len(sel), 200)])
    t = torch.from_numpy(sel).cuda()
    assert torch.equal(ids[t], s_ids) and torch.equal(dists[t], s_d)


@pytest.mark.parametrize("tiles", [2, 3, 4])
@pytest.mark.parametrize("N,Nq,D", [
    (300, 100_000, 160),   # 782 query blocks of 2-5 units each: every workgroup runs several segments
    (1000, 60_000, 200),   # ranges a bit longer than one query block
    (2100, 40_000, 132),   # a range = a fraction of a query block, not a multiple of its units
])
def test_bf_mfma_chunked_many_segments_per_workgroup(ops, N, Nq, D, tiles):
    """Chunked kernel (D > 128, T = 2 / 3 / 4 tiles per accumulator group) with ranges that cross
    many query-block boundaries: a workgroup runs several segments in one do/while, and a wave that
    starts the next segment early must not disturb what a slower wave of the same workgroup still
    reads in the previous segment's epilogue (the row norms `bn_grp`; round-4 advisor finding: the
    next segment's threshold initialisation aliased them).  Fractional data, K = 10: every query
    against the scan kernel would be slow, a strided sample + the block borders are compared."""
    from ggnn_amd import _lib
    rng = np.random.default_rng(77)
    base = rng.normal(size=(N, D)).astype(np.float32)
    q = rng.normal(size=(Nq, D)).astype(np.float32)
    b, qq = dev(base), dev(q)
    with _lib.hooks(BF_TILES=tiles):
        for rep in range(3):   # timing-dependent: a few launches
            ids, dists = ops.bf_query(b, qq, 10, 0)
            if rep == 0:
                ids0, dists0 = ids.clone(), dists.clone()
            else:
                assert torch.equal(ids, ids0) and torch.equal(dists, dists0)
    sel = np.unique(np.concatenate([np.arange(0, Nq, max(1, Nq // 1500)), np.arange(Nq - 130, Nq),
                                    np.arange(120, 140), np.arange(128 * 37 - 5, 128 * 37 + 5)]))
    sub = dev(q[sel])
    parts = [ops.bf_query(b, sub[i:i + 200].contiguous(), 10, 0) for i in range(0, len(sel), 200)]
    t = torch.from_numpy(sel).cuda()
    assert torch.equal(ids[t], torch.cat([p[0] for p in parts]))
    assert torch.equal(dists[t], torch.cat([p[1] for p in parts]))


@pytest.mark.parametrize("N,D,Nq,K", [(50_000, 128, 700, 10), (33_333, 128, 257, 3), (20_001, 96, 300, 16),
                                      (9_000, 64, 513, 10), (4_100, 32, 256, 4), (70_000, 128, 1000, 1)])
def test_uint8_register_list_kernel_exact(orc, monkeypatch, N, D, Nq, K):
    """bf_i8v2_kernel (uint8, squared L2, K <= 16: per-query sets in registers, thresholds folded
    into the accumulators and shared between the slices through atomics): results must be the
    oracle's bit for bit -- with heavy ties (duplicated rows, extreme bytes), ragged sizes, many
    slices, with and without the seeding launch, and equal to the LDS-list kernel it replaces."""
    from ggnn_amd import ops
    rng = np.random.default_rng(N + D + K)
    base = rng.integers(0, 256, (N, D)).astype(np.uint8)
    base[::97] = 255
    base[5::101] = 0
    base[N // 2:N // 2 + 400] = base[:400]          # ties across slices: lower index first
    base[N - 300:] = base[100:400]                  # ... and at the very end of the base
    q = rng.integers(0, 256, (Nq, D)).astype(np.uint8)
    q[3] = 0
    q[4] = 255
    q[5:205] = base[100:300]                        # exact hits with three copies each
    o_ids, o_d = orc.bf_query(base, q, K)
    d_base, d_q = torch.from_numpy(base).cuda(), torch.from_numpy(q).cuda()
    # slices: the bound exchange between them (bf_i8.hip) with the default (BF_I8_RANKS = -1: ONE
    # position chosen from the slice count), ALL five positions together (31: the exchange with
    # m = 10, 5, 4, 2, 1 at once), single positions (16 = the last entry alone: rounds 3-4), too few
    # slices for most positions, 32 slices, none
    for env in ({}, {"GGNN_BF_SLICES": "7"}, {"GGNN_BF_SLICES": "12", "GGNN_BF_I8_RANKS": "31"},
                {"GGNN_BF_SLICES": "7", "GGNN_BF_I8_RANKS": "31"}, {"GGNN_BF_SLICES": "23", "GGNN_BF_I8_RANKS": "16"},
                {"GGNN_BF_SLICES": "1"}, {"GGNN_BF_I8_V1": "1"}, {"GGNN_BF_SLICES": "2"},
                {"GGNN_BF_SLICES": "5", "GGNN_BF_I8_RANKS": "3"}, {"GGNN_BF_SLICES": "32"},
                {"GGNN_BF_SLICES": "12", "GGNN_BF_I8_RANKS": "2"},
                {"GGNN_BF_SLICES": "12", "GGNN_BF_I8_NOSHARE": "1"},
                {"GGNN_BF_SLICES": "12", "GGNN_BF_I8_SEED": "0"},
                {"GGNN_BF_SLICES": "9", "GGNN_BF_I8_SEED": "4096", "GGNN_BF_I8_REFRESH": "2"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        ids, d = ops.bf_query(d_base, d_q, K)
        for k_ in env:
            monkeypatch.delenv(k_)
        assert np.array_equal(ids.cpu().numpy(), o_ids), env
        assert np.array_equal(d.cpu().numpy(), o_d), env


@pytest.mark.parametrize("scale", [1e-19, 3e-21, 1e-23])
def test_tiny_magnitudes_stay_exact(ops, scale):
    """coordinates so small that squares and products are denormal (or flush to
    zero).  The certificate's rounding model is relative; the absolute error of the denormal
    range is covered by an explicit term, so such queries are either still certified correctly
    or re-scanned -- the answer must equal the scan kernel's in every case."""
    N, D, Nq, K = 20_000, 128, 300, 10
    base = (_clustered(N, D, 41).astype(np.float64) * scale).astype(np.float32)
    q = (_clustered(Nq, D, 42).astype(np.float64) * scale).astype(np.float32)
    b, qq = dev(base), dev(q)
    ids, d = ops.bf_query(b, qq, K, 0)
    s_ids, s_d = scan_answer(ops, b, qq, K, 0)
    assert torch.equal(d, s_d)
    # distances that underflow to equal values may order tied rows either way: compare as sets
    # wherever the distances at the boundary tie, exactly otherwise
    same = (ids == s_ids).all(1)
    for n in torch.nonzero(~same).flatten().tolist():
        assert set(ids[n].tolist()) == set(s_ids[n].tolist()) or float(d[n, -1]) == float(d[n, 0]), n
