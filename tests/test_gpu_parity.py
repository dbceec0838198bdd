"""GPU parity tests: every HIP kernel against the CPU oracle on the same seeded inputs, through
the C-ABI (ggnn_amd.ops -> ggnn_op_*).  Integer-valued inputs => bit-exact ids AND distances;
uniform float inputs => 1e-4 relative on distances (tolerance from BASELINE.json north_star)."""
import numpy as np
import pytest

from conftest import make_int_data, make_uni_data
from parity_helpers import assert_topk_parity, cos_atol

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

RTOL = 1e-4


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from ggnn_amd import ops as o
    return o


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def start_points(g):
    c = g["cfg"]
    return g["tr"][c.STs_offsets[3]:c.STs_offsets[3] + c.Ns[3]]


# ---------------------------------------------------------------------------------------------
# bf_query
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,D,Nq,K", [(3000, 128, 64, 10), (1000, 96, 33, 100), (517, 32, 5, 7),
                                      (5000, 960, 16, 10), (70000, 128, 8, 10)])
def test_bf_query_int_exact(ops, orc, N, D, Nq, K):
    base, q = make_int_data(N, D, 1), make_int_data(Nq, D, 2)
    ids, d = ops.bf_query(dev(base), dev(q), K)
    o_ids, o_d = orc.bf_query(base, q, K)
    assert np.array_equal(ids.cpu().numpy(), o_ids)
    assert np.array_equal(d.cpu().numpy(), o_d)


@pytest.mark.parametrize("N,D,Nq,K", [(3000, 32, 9, 300), (1500, 16, 3, 1000), (50000, 32, 4, 257)])
def test_bf_query_very_large_k(ops, orc, N, D, Nq, K):
    base, q = make_int_data(N, D, 13), make_int_data(Nq, D, 14)
    ids, d = ops.bf_query(dev(base), dev(q), K)
    o_ids, o_d = orc.bf_query(base, q, K)
    assert np.array_equal(ids.cpu().numpy(), o_ids)
    assert np.array_equal(d.cpu().numpy(), o_d)


def test_bf_query_ties_lower_index_first(ops, orc):
    # duplicated base rows: equal distances must keep the lower base index first (Q2)
    base = make_int_data(500, 64, 3)
    base = np.concatenate([base, base, base])
    q = make_int_data(20, 64, 4)
    ids, d = ops.bf_query(dev(base), dev(q), 12)
    o_ids, o_d = orc.bf_query(base, q, 12)
    assert np.array_equal(ids.cpu().numpy(), o_ids)
    assert np.array_equal(d.cpu().numpy(), o_d)


def test_bf_query_fewer_points_than_k(ops, orc):
    base, q = make_int_data(6, 32, 5), make_int_data(3, 32, 6)
    ids, d = ops.bf_query(dev(base), dev(q), 10)
    o_ids, o_d = orc.bf_query(base, q, 10)
    assert np.array_equal(ids.cpu().numpy(), o_ids)
    assert np.array_equal(d.cpu().numpy(), o_d)
    assert (ids.cpu().numpy()[:, 6:] == -1).all() and np.isinf(d.cpu().numpy()[:, 6:]).all()


@pytest.mark.parametrize("measure", [0, 1])
def test_bf_query_float_tolerance(ops, orc, measure):
    base, q = make_uni_data(4000, 128, 7), make_uni_data(50, 128, 8)
    ids, d = ops.bf_query(dev(base), dev(q), 10, measure)
    o_ids, o_d = orc.bf_query(base, q, 10, measure)
    np.testing.assert_allclose(d.cpu().numpy(), o_d, rtol=RTOL, atol=cos_atol(128) if measure else 0)
    # float64 truth: every distance within 1e-4, every point a top-K point, and ids differ from
    # the oracle's only where the two candidates are a near-tie
    assert_topk_parity(base, q, ids.cpu().numpy(), d.cpu().numpy(), o_ids, 10, measure)


def test_bf_query_uint8(ops, orc):
    base = np.random.default_rng(9).integers(0, 256, (3000, 128)).astype(np.uint8)
    q = np.random.default_rng(10).integers(0, 256, (40, 128)).astype(np.uint8)
    ids, d = ops.bf_query(dev(base), dev(q), 10)
    o_ids, o_d = orc.bf_query(base, q, 10)
    assert np.array_equal(ids.cpu().numpy(), o_ids)
    assert np.array_equal(d.cpu().numpy(), o_d)


@pytest.mark.parametrize("N,D,Nq,K", [(9000, 128, 300, 10), (5000, 64, 257, 10), (7000, 96, 300, 24),
                                      (4100, 32, 260, 5), (6000, 112, 300, 100), (40000, 128, 512, 10)])
def test_bf_query_uint8_matrix_path(ops, orc, N, D, Nq, K):
    """uint8 + L2 with >= 256 queries and >= 4096 rows runs on v_mfma_i32_32x32x32_i8 (bytes
    shifted to signed); extreme byte values and duplicated rows included"""
    rng = np.random.default_rng(N + D)
    base = rng.integers(0, 256, (N, D)).astype(np.uint8)
    base[::97] = 255
    base[5::101] = 0
    base[N // 2:N // 2 + 50] = base[:50]  # ties: the lower index comes first
    q = rng.integers(0, 256, (Nq, D)).astype(np.uint8)
    q[3] = 0
    q[4] = 255
    q[5] = base[7]
    ids, d = ops.bf_query(dev(base), dev(q), K)
    o_ids, o_d = orc.bf_query(base, q, K)
    assert np.array_equal(ids.cpu().numpy(), o_ids)
    assert np.array_equal(d.cpu().numpy(), o_d)


# ---------------------------------------------------------------------------------------------
# query
# ---------------------------------------------------------------------------------------------
# the last two cases pop more keys than the visited ring holds (cache 256, ring 192): entries are
# overwritten in ring order and leave the visited hash set again
@pytest.mark.parametrize("K,tau,iters", [(10, 0.34, 200), (10, 0.64, 400), (1, 0.5, 100),
                                         (40, 0.6, 400), (100, 0.6, 512), (10, 0.9, 1000),
                                         (10, 2.5, 250), (24, 3.0, 255), (10, 3.0, 512),
                                         (40, 3.0, 500)])
def test_query_int_exact(ops, orc, small_graph, K, tau, iters):
    g = small_graph
    q = make_int_data(128, g["D"], 4321)
    graph0 = g["graph"][:g["N"]]
    ids, d, nd, npop = ops.query(dev(g["base"]), dev(q), dev(graph0), dev(start_points(g)),
                                 dev(g["stats"]), K, tau, iters, counters=True)
    o_ids, o_d, o_nd, o_np = orc.query(g["base"], q, graph0, start_points(g), g["stats"], K, tau,
                                       iters, counters=True)
    assert np.array_equal(ids.cpu().numpy(), o_ids)
    assert np.array_equal(d.cpu().numpy(), o_d)
    assert np.array_equal(npop.cpu().numpy().astype(np.uint32), o_np)
    assert np.array_equal(nd.cpu().numpy().astype(np.uint32), o_nd)
    if iters in (250, 255):
        assert int(o_np.max()) > 192 + 16, "the case is meant to wrap the visited ring"
    if iters in (500, 512):
        assert int(o_np.max()) > 480 + 16, "the case is meant to wrap the 480-entry ring"
    # 1000..2048 iterations: rings of 992 / 2016 keys, searched by the ring scan (a hashed set with
    # four / eight bucket registers was exact too but slower: its LDS cost occupancy)
    if (tau, iters) == (4.0, 1000):
        assert int(o_np.max()) > 900, "the case is meant to fill the 992-entry ring"
    if (tau, iters) == (5.0, 2048):
        assert int(o_np.max()) > 1500, "the case is meant to go deep into the 2016-entry ring"


# The visited ring is mirrored in a hash set (traversal.hpp, SortedList<R, HB>): buckets of 8 keys,
# a stash for keys whose bucket is full, the ring scan when the stash overflows, removal when the
# ring wraps.  GGNN_VIS_SLOTS shrinks the buckets so that ordinary searches take all of these paths:
# 1-2 slots overflow the stash (ring scan takes over), 4 slots fill it without overflowing (keys
# are removed from buckets AND stash when the ring wraps).  Results and counters must not change.
@pytest.mark.parametrize("slots", [1, 2, 4])
@pytest.mark.parametrize("K,tau,iters", [(10, 0.64, 200), (10, 2.5, 250), (24, 3.0, 255),
                                         (10, 0.64, 400), (10, 3.0, 512), (40, 3.0, 500),
                                         (10, 4.0, 1000), (10, 0.8, 1024), (10, 5.0, 2048),
                                         (10, 1.0, 1500), (10, 3.0, 480), (40, 3.0, 448),
                                         (10, 3.0, 300)])
def test_query_visited_hash_paths_exact(ops, orc, small_graph, slots, K, tau, iters, monkeypatch):
    g = small_graph
    q = make_int_data(96, g["D"], 4322)
    graph0 = g["graph"][:g["N"]]
    o_ids, o_d, o_nd, o_np = orc.query(g["base"], q, graph0, start_points(g), g["stats"], K, tau,
                                       iters, counters=True)
    monkeypatch.setenv("GGNN_VIS_SLOTS", str(slots))
    # the pre-screened float kernel is the one that carries the hash set (the plain two-chunk
    # float kernel keeps the ring scan, see launch_query_r)
    b = dev(g["base"])
    ps = ops.prescreen_encode(b)
    ids, d, nd, npop = ops.query(b, dev(q), dev(graph0), dev(start_points(g)), dev(g["stats"]), K,
                                 tau, iters, counters=True, prescreen=ps)
    assert np.array_equal(ids.cpu().numpy(), o_ids)
    assert np.array_equal(d.cpu().numpy(), o_d)
    assert np.array_equal(npop.cpu().numpy().astype(np.uint32), o_np)
    assert np.array_equal(nd.cpu().numpy().astype(np.uint32), o_nd)
    # (and the kernels of launches without counters: stash / overflow list / ring scan as above)
    ids, d = ops.query(b, dev(q), dev(graph0), dev(start_points(g)), dev(g["stats"]), K, tau, iters,
                       prescreen=ps)
    assert np.array_equal(ids.cpu().numpy(), o_ids) and np.array_equal(d.cpu().numpy(), o_d)
    if iters in (250, 255):
        assert int(o_np.max()) > 192 + 16, "the case is meant to wrap the 192-entry ring"
    if iters in (500, 512):
        assert int(o_np.max()) > 480 + 16, "the case is meant to wrap the 480-entry ring"
    # 257..480 iterations that cannot wrap their ring (480 / 448 keys): the early-rows kernel has NO
    # ring (SortedList<1, 2, true>: buckets + stash are the visited keys); with 1-2 usable slots
    # the stash overflows and later keys go to the overflow list in global memory, which every
    # membership test then scans
    if iters in (480, 448):
        assert int(o_np.max()) > 400, "the case is meant to fill most of its ring"
    # 1000..2048 iterations: rings of 992 / 2016 keys kept in global memory and mirrored in the
    # 16-bit tag set (traversal.hpp "long rings": 7 tags + their number per 16-byte bucket): 1-2
    # usable slots overflow the stash (the scan of the global ring takes over), 4 fill it; 1000 /
    # 1024 iterations on a 992-key ring wrap it
    if (tau, iters) == (4.0, 1000):
        assert int(o_np.max()) > 900, "the case is meant to fill the 992-entry ring"
    if (tau, iters) == (5.0, 2048):
        assert int(o_np.max()) > 1500, "the case is meant to go deep into the 2016-entry ring"


@pytest.mark.parametrize("dtype", ["f32", "u8"])
@pytest.mark.parametrize("K,tau,iters", [(10, 0.64, 175), (10, 0.9, 400), (40, 2.0, 448),
                                         (10, 3.0, 256), (10, 1.0, 480), (10, 3.0, 600),
                                         (40, 2.0, 960), (10, 3.0, 1900)])
def test_query_orders_and_ring_homes_equal_the_oracle(ops, orc, small_graph, dtype, K, tau, iters):
    """The query kernel's order variants (hook QUERY_EARLY: first-read rows of a pop's neighbours
    requested before / after the pop's bookkeeping and the membership test; hook
    QUERY_GLOBAL_RING: visited ring of a 512-key cache in global memory / in LDS) against the
    oracle: ids, distances, n_dist, n_pop -- float32 with the pre-screen and uint8 rows (the two
    early-rows layouts).  600 / 960 / 1900 iterations: the 16-bit tag set of long rings (992 / 960 /
    2016 keys), ring-less when the search cannot wrap."""
    from ggnn_amd import _lib
    g = small_graph
    base = g["base"] if dtype == "f32" else g["base"].astype(np.uint8)
    q = make_int_data(80, g["D"], 4324)
    q = q if dtype == "f32" else q.astype(np.uint8)
    graph0 = g["graph"][:g["N"]]
    o_ids, o_d, o_nd, o_np = orc.query(base, q, graph0, start_points(g), g["stats"], K, tau, iters,
                                       counters=True)
    b = dev(base)
    ps = ops.prescreen_encode(b) if dtype == "f32" else None
    for early, gring in ((1, 1), (1, 0), (0, 1)):
        with _lib.hooks(QUERY_EARLY=early, QUERY_GLOBAL_RING=gring):
            ids, d, nd, npop = ops.query(b, dev(q), dev(graph0), dev(start_points(g)),
                                         dev(g["stats"]), K, tau, iters, counters=True, prescreen=ps)
        assert np.array_equal(ids.cpu().numpy(), o_ids), (early, gring)
        assert np.array_equal(d.cpu().numpy(), o_d), (early, gring)
        assert np.array_equal(npop.cpu().numpy().astype(np.uint32), o_np), (early, gring)
        assert np.array_equal(nd.cpu().numpy().astype(np.uint32), o_nd), (early, gring)
        # the same launch WITHOUT counters: the ring-less kernels then test the sorted part of the
        # cache behind the verdicts, for the candidates still in the race (fetch_early<.., false>)
        with _lib.hooks(QUERY_EARLY=early, QUERY_GLOBAL_RING=gring):
            ids, d = ops.query(b, dev(q), dev(graph0), dev(start_points(g)), dev(g["stats"]), K, tau,
                               iters, prescreen=ps)
        assert np.array_equal(ids.cpu().numpy(), o_ids), (early, gring, "no counters")
        assert np.array_equal(d.cpu().numpy(), o_d), (early, gring, "no counters")


@pytest.mark.parametrize("dtype", ["f32", "u8"])
@pytest.mark.parametrize("KB", [1, 3, 8, 17, 23, 24])
def test_query_early_rows_on_arbitrary_graph_rows(ops, orc, dtype, KB):
    """The early-rows order requests the rows of ALL slots of a graph row before it knows which of
    them are candidates at all.  Hand-made graphs (not built ones): rows with EMPTY (-1) slots,
    the same neighbour several times in a row (the reference evaluates both copies: the membership
    test runs before either is pushed), self loops, and every row width up to 24 -- ids, distances,
    n_dist and n_pop must equal the oracle's in both orders."""
    from ggnn_amd import _lib
    N, D = 1500, 128
    r = np.random.default_rng(700 + KB)
    base = r.integers(0, 256, (N, D))
    base = base.astype(np.uint8) if dtype == "u8" else base.astype(np.float32)
    q = r.integers(0, 256, (60, D))
    q = q.astype(np.uint8) if dtype == "u8" else q.astype(np.float32)
    graph = r.integers(0, N, (N, KB)).astype(np.int32)
    graph[r.random((N, KB)) < 0.15] = -1                       # EMPTY slots anywhere in a row
    dup = r.random(N) < 0.3
    if KB >= 3:
        graph[dup, KB - 1] = graph[dup, 0]                      # a neighbour twice in one row
    graph[::7, 0] = np.arange(0, N, 7)                          # self loops
    graph[5] = -1                                               # a row without any neighbour
    start = r.choice(N, 32, replace=False).astype(np.int32)
    stats = np.array([900.0, 1400.0], np.float32)
    o_ids, o_d, o_nd, o_np = orc.query(base, q, graph, start, stats, 10, 1.2, 150, counters=True)
    b = dev(base)
    ps = ops.prescreen_encode(b) if dtype == "f32" else None
    for early in (1, 0):
        with _lib.hooks(QUERY_EARLY=early):
            ids, d, nd, npop = ops.query(b, dev(q), dev(graph), dev(start), dev(stats), 10, 1.2, 150,
                                         counters=True, prescreen=ps)
        assert np.array_equal(ids.cpu().numpy(), o_ids), early
        assert np.array_equal(d.cpu().numpy(), o_d), early
        assert np.array_equal(nd.cpu().numpy().astype(np.uint32), o_nd), early
        assert np.array_equal(npop.cpu().numpy().astype(np.uint32), o_np), early
    # without counters (sorted part tested behind the verdicts): duplicates in a row, self loops and
    # EMPTY slots must give the same lists
    ids, d = ops.query(b, dev(q), dev(graph), dev(start), dev(stats), 10, 1.2, 150, prescreen=ps)
    assert np.array_equal(ids.cpu().numpy(), o_ids) and np.array_equal(d.cpu().numpy(), o_d)
    assert int(o_np.max()) > 20


@pytest.mark.parametrize("top,btm", [(3, 0), (2, 0), (2, 1)])
def test_merge_orders_equal_the_oracle(ops, orc, small_graph, top, btm):
    """hook MERGE_EARLY = 1 | 0 (the same reordering in the merge kernel, upper layers go through
    the translation) with and without the pre-screen"""
    from ggnn_amd import _lib
    g = small_graph
    c = g["cfg"]
    o_gb, o_nn1 = orc.merge(g["base"], c, g["graph"], g["tr"], g["sel"], g["stats"], 0.5, top, btm)
    b = dev(g["base"])
    for early in (1, 0):
        for ps in (None, ops.prescreen_encode(b)):
            with _lib.hooks(MERGE_EARLY=early):
                gb, nn1 = ops.merge(b, c, dev(g["graph"]), dev(g["tr"]), dev(g["sel"]),
                                    dev(g["stats"]), 0.5, top, btm, prescreen=ps)
            assert np.array_equal(gb.cpu().numpy(), o_gb), (early, ps is not None)
            if btm == 0:
                assert np.array_equal(nn1.cpu().numpy(), o_nn1), (early, ps is not None)


@pytest.mark.parametrize("K,tau,iters", [(10, 4.0, 1000), (10, 6.0, 1024), (30, 5.0, 2048),
                                         (10, 1.0, 1500), (47, 3.0, 700)])
def test_query_long_ring_tag_set_equals_ring_scan(ops, orc, small_graph, K, tau, iters):
    """Rings of 992 / 2016 keys: the tag set (default) and the ring scan (hook VIS_TAG_SET = 0)
    give the oracle's ids, distances and counters; (6.0, 1024) runs past the 992-key ring, so the
    last pops of the tag-set kernel scan the wrapped ring in global memory."""
    from ggnn_amd import _lib
    g = small_graph
    q = make_int_data(64, g["D"], 4323)
    graph0 = g["graph"][:g["N"]]
    o_ids, o_d, o_nd, o_np = orc.query(g["base"], q, graph0, start_points(g), g["stats"], K, tau,
                                       iters, counters=True)
    b = dev(g["base"])
    ps = ops.prescreen_encode(b)
    for tag_set in (1, 0):
        with _lib.hooks(VIS_TAG_SET=tag_set):
            ids, d, nd, npop = ops.query(b, dev(q), dev(graph0), dev(start_points(g)),
                                         dev(g["stats"]), K, tau, iters, counters=True, prescreen=ps)
        assert np.array_equal(ids.cpu().numpy(), o_ids), tag_set
        assert np.array_equal(d.cpu().numpy(), o_d), tag_set
        assert np.array_equal(npop.cpu().numpy().astype(np.uint32), o_np), tag_set
        assert np.array_equal(nd.cpu().numpy().astype(np.uint32), o_nd), tag_set
    if iters == 1024:
        assert int(o_np.max()) > 992, "the case is meant to wrap the 992-entry ring"


@pytest.mark.parametrize("slots", [1, 4])
@pytest.mark.parametrize("top,btm", [(3, 0), (2, 1)])
def test_merge_visited_hash_paths_exact(ops, orc, small_graph, slots, top, btm, monkeypatch):
    g = small_graph
    c = g["cfg"]
    o_gb, o_nn1, o_nd = orc.merge(g["base"], c, g["graph"], g["tr"], g["sel"], g["stats"], 0.5,
                                  top, btm, counters=True)
    monkeypatch.setenv("GGNN_VIS_SLOTS", str(slots))
    b = dev(g["base"])
    gb, nn1, nd = ops.merge(b, c, dev(g["graph"]), dev(g["tr"]), dev(g["sel"]), dev(g["stats"]),
                            0.5, top, btm, counters=True, prescreen=ops.prescreen_encode(b))
    assert np.array_equal(gb.cpu().numpy(), o_gb)
    assert np.array_equal(nd.cpu().numpy().astype(np.uint32), o_nd)
    gb2, _ = ops.merge(b, c, dev(g["graph"]), dev(g["tr"]), dev(g["sel"]), dev(g["stats"]), 0.5, top,
                       btm, prescreen=ops.prescreen_encode(b))     # (the non-counting kernel)
    assert np.array_equal(gb2.cpu().numpy(), o_gb)


def test_query_shard_offsets(ops, orc, small_graph):
    g = small_graph
    q = make_int_data(32, g["D"], 11)
    graph0 = g["graph"][:g["N"]]
    out_i = torch.full((32, 30), -5, dtype=torch.int32, device="cuda")
    out_d = torch.full((32, 30), -5.0, dtype=torch.float32, device="cuda")
    ops.query(dev(g["base"]), dev(q), dev(graph0), dev(start_points(g)), dev(g["stats"]), 10, 0.5,
              200, shards_per_gpu=3, on_gpu_shard=2, out=(out_i, out_d))
    o = (np.full((32, 30), -5, np.int32), np.full((32, 30), -5.0, np.float32))
    orc.query(g["base"], q, graph0, start_points(g), g["stats"], 10, 0.5, 200, shards_per_gpu=3,
              on_gpu_shard=2, out=o)
    assert np.array_equal(out_i.cpu().numpy(), o[0])
    assert np.array_equal(out_d.cpu().numpy(), o[1])


def test_query_float_tolerance(ops, orc):
    N, D, K = 1500, 64, 24
    base = make_uni_data(N, D, 21)
    cfg, graph, tr, sel, stats = orc.build(base, K, 0.5, 0, rng=orc.make_rng(N, 5))
    q = make_uni_data(64, D, 22)
    start = tr[cfg.STs_offsets[3]:cfg.STs_offsets[3] + cfg.Ns[3]]
    ids, d, nd, npop = ops.query(dev(base), dev(q), dev(graph[:N]), dev(start), dev(stats), 10, 0.6,
                                 400, counters=True)
    # fractional float32 data: with the oracle summing in the kernels' order every decision of
    # the search is the same decision on both sides -- ids, distances and counters bit for bit
    with orc.wave_order():
        o_ids, o_d, o_nd, o_np = orc.query(base, q, graph[:N], start, stats, 10, 0.6, 400,
                                           counters=True)
    assert np.array_equal(ids.cpu().numpy(), o_ids)
    assert np.array_equal(d.cpu().numpy(), o_d)
    assert np.array_equal(nd.cpu().numpy().astype(np.uint32), o_nd)
    assert np.array_equal(npop.cpu().numpy().astype(np.uint32), o_np)
    # the reference's own (restated) cub::BlockReduce order: distances of common ids within 1e-4
    r_ids, r_d = orc.query(base, q, graph[:N], start, stats, 10, 0.6, 400)
    same = r_ids == o_ids
    np.testing.assert_allclose(o_d[same], r_d[same], rtol=RTOL)


def test_query_cosine(ops, orc):
    N, D, K = 1500, 64, 24
    base = make_int_data(N, D, 31)
    cfg, graph, tr, sel, stats = orc.build(base, K, 0.5, 0, measure=1, rng=orc.make_rng(N, 6))
    q = make_int_data(64, D, 32)
    start = tr[cfg.STs_offsets[3]:cfg.STs_offsets[3] + cfg.Ns[3]]
    ids, d = ops.query(dev(base), dev(q), dev(graph[:N]), dev(start), dev(stats), 10, 0.6, 400, 1)
    with orc.wave_order():
        o_ids, o_d = orc.query(base, q, graph[:N], start, stats, 10, 0.6, 400, 1)
    assert np.array_equal(ids.cpu().numpy(), o_ids)
    assert np.array_equal(d.cpu().numpy(), o_d)
    r_ids, r_d = orc.query(base, q, graph[:N], start, stats, 10, 0.6, 400, 1)
    same = r_ids == o_ids
    np.testing.assert_allclose(o_d[same], r_d[same], rtol=RTOL, atol=cos_atol(D))


# ---------------------------------------------------------------------------------------------
# construction kernels with injected inputs
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("layer", [0, 1, 3])
def test_top_exact(ops, orc, small_graph, layer):
    g = small_graph
    c = g["cfg"]
    tr_l = None if layer == 0 else g["tr"][c.STs_offsets[layer]:c.STs_offsets[layer] + c.Ns[layer]]
    S, S_off = (c.S0, c.S0_off) if layer == 0 else (c.S, 0)
    graph, nn1 = ops.top(dev(g["base"]), g["K"], None if tr_l is None else dev(tr_l),
                         c.Ns[layer], S, S_off, layer)
    o_graph, o_nn1 = orc.top(g["base"], g["K"], tr_l, c.Ns[layer], S, S_off, layer)
    assert np.array_equal(graph.cpu().numpy(), o_graph)
    assert np.array_equal(nn1.cpu().numpy(), o_nn1)


def test_select_exact(ops, orc, small_graph):
    g = small_graph
    c = g["cfg"]
    nn1 = np.random.default_rng(3).random(c.N, dtype=np.float32) * 100 + 1
    rng = orc.make_rng(c.N, 17)[0]
    for layer in (0, 1, 2):
        tr, sel = g["tr"].copy(), g["sel"].copy()
        d_tr, d_sel = dev(tr), dev(sel)
        ops.select(c, layer, dev(nn1), dev(rng), d_tr, d_sel)
        orc.select(c, layer, nn1, rng, tr, sel)
        assert np.array_equal(d_tr.cpu().numpy(), tr)
        assert np.array_equal(d_sel.cpu().numpy(), sel)


@pytest.mark.parametrize("top,btm", [(1, 0), (2, 0), (3, 0), (3, 1), (3, 2), (2, 1)])
def test_merge_exact(ops, orc, small_graph, top, btm):
    g = small_graph
    c = g["cfg"]
    gb, nn1, nd = ops.merge(dev(g["base"]), c, dev(g["graph"]), dev(g["tr"]), dev(g["sel"]),
                            dev(g["stats"]), 0.5, top, btm, counters=True)
    o_gb, o_nn1, o_nd = orc.merge(g["base"], c, g["graph"], g["tr"], g["sel"], g["stats"], 0.5,
                                  top, btm, counters=True)
    assert np.array_equal(gb.cpu().numpy(), o_gb)
    assert np.array_equal(nd.cpu().numpy().astype(np.uint32), o_nd)
    if btm == 0:
        assert np.array_equal(nn1.cpu().numpy(), o_nn1)


@pytest.mark.parametrize("layer", [0, 2])
def test_sym_sequential_exact(ops, orc, small_graph, layer):
    """sym is racy by design (atomics + cross-block reads of sym_buffer); launched one block at
    a time in ascending order it is deterministic and must equal the oracle's serialisation."""
    g = small_graph
    c = g["cfg"]
    K, KF = g["K"], g["K"] // 2
    Nl = min(c.Ns[layer], 300)
    graph_l = g["graph"][c.Ns_offsets[layer]:c.Ns_offsets[layer] + c.Ns[layer]].copy()
    tr_l = None if layer == 0 else g["tr"][c.STs_offsets[layer]:c.STs_offsets[layer] + c.Ns[layer]]
    sb = np.full((c.Ns[layer], KF), -1, np.int32)
    sa = np.zeros(c.Ns[layer], np.uint32)
    orc.margin_reset()
    orc.sym(g["base"], K, graph_l, tr_l, g["stats"], 0.5, sb, sa, first_n=0, count=Nl)
    assert orc.margin_min() > 1e-5, "seed is not decision tie-free for the half-point test"
    d_sb = torch.full((c.Ns[layer], KF), -1, dtype=torch.int32, device="cuda")
    d_sa = torch.zeros(c.Ns[layer], dtype=torch.int32, device="cuda")
    d_base, d_graph, d_stats = dev(g["base"]), dev(graph_l), dev(g["stats"])
    d_tr = None if tr_l is None else dev(tr_l)
    for n in range(Nl):
        ops.sym(d_base, K, d_graph, d_tr, d_stats, 0.5, d_sb, d_sa, first_n=n, count=1)
    assert np.array_equal(d_sa.cpu().numpy().astype(np.uint32), sa)
    assert np.array_equal(d_sb.cpu().numpy(), sb)


def test_sym_buffer_merge_exact(ops, orc, small_graph):
    g = small_graph
    c = g["cfg"]
    K, KF = g["K"], g["K"] // 2
    N = c.N
    r = np.random.default_rng(8)
    graph_l = g["graph"][:N].copy()
    sa = r.integers(0, KF + 4, N).astype(np.uint32)
    sb = np.full((N, KF), -1, np.int32)
    for n in range(N):
        cnt = min(int(sa[n]), KF)
        sb[n, :cnt] = r.choice(N, cnt, replace=False)
        if cnt and r.random() < 0.5:  # make some requested links collide with existing ones
            sb[n, 0] = graph_l[n, K - KF + r.integers(0, KF)]
    d_graph = dev(graph_l)
    ops.sym_buffer_merge(K, dev(sb), dev(sa.astype(np.int32)), d_graph)
    orc.sym_buffer_merge(K, sb, sa, graph_l)
    assert np.array_equal(d_graph.cpu().numpy(), graph_l)


def test_nn1_stats(ops, orc):
    v = (np.random.default_rng(12).random(100003, dtype=np.float32) * 500).astype(np.float32)
    out = ops.nn1_stats(dev(v)).cpu().numpy()
    o = orc.nn1_stats(v)
    assert out[1] == o[1] == v.max()
    np.testing.assert_allclose(out[0], o[0], rtol=1e-5)
    np.testing.assert_allclose(out[0], v.astype(np.float64).mean(), rtol=1e-5)


def test_uniform_range(ops):
    u = ops.uniform(100000, 1234, 3).cpu().numpy()
    assert u.min() > 0.0 and u.max() <= 1.0
    assert abs(u.mean() - 0.5) < 0.01
    u2 = ops.uniform(100000, 1234, 4).cpu().numpy()
    assert not np.array_equal(u, u2)


# ---------------------------------------------------------------------------------------------
# shard results
# ---------------------------------------------------------------------------------------------
def test_sort_and_merge_results(ops, orc):
    r = np.random.default_rng(13)
    Nq, K, shards = 200, 10, 4
    # tie-free distances (distinct integers)
    d = r.permutation(Nq * K * shards * 3)[:Nq * K * shards].reshape(Nq, K * shards)
    d = d.astype(np.float32)
    ids = r.integers(0, 1000, (Nq, K * shards)).astype(np.int32)
    # each shard's K block is sorted, as the query kernel writes it
    for s in range(shards):
        blk = slice(s * K, (s + 1) * K)
        order = np.argsort(d[:, blk], 1)
        d[:, blk] = np.take_along_axis(d[:, blk], order, 1)
        ids[:, blk] = np.take_along_axis(ids[:, blk], order, 1)
    di, dd = dev(ids), dev(d)
    ops.sort_shard_results(di, dd)
    oi, od = orc.sort_shard_results(ids, d)
    assert np.array_equal(di.cpu().numpy(), oi) and np.array_equal(dd.cpu().numpy(), od)

    # multi-GPU merge: parts = per-GPU sorted rows
    G, N_shard = 3, 1000
    parts_i = [r.integers(0, N_shard, (Nq, K)).astype(np.int32) for _ in range(G)]
    alld = r.permutation(Nq * K * G * 2)[:Nq * K * G].astype(np.float32).reshape(G, Nq, K)
    parts_d = [np.sort(alld[gidx], 1) for gidx in range(G)]
    mi, md = ops.merge_results(dev(np.stack(parts_i)), dev(np.stack(parts_d)), K, N_shard)
    oi, od = orc.merge_results(parts_i, parts_d, K, 1, N_shard)
    assert np.array_equal(mi.cpu().numpy(), oi) and np.array_equal(md.cpu().numpy(), od)


# ---------------------------------------------------------------------------------------------
# end to end through the ggnn surface
# ---------------------------------------------------------------------------------------------
def test_end_to_end_recall_and_query_parity(orc):
    import ggnn_amd as ggnn
    N, D, K = 6000, 128, 10
    base, q = make_int_data(N, D, 41), make_int_data(200, D, 42)
    eng = ggnn.GGNN()
    eng.set_base(base)
    eng.build(24, 0.5, 2)
    ids, d = eng.query(q, K, 0.64, 400)
    gt, gtd = eng.bf_query(q, K)
    assert ids.dtype == torch.int32 and d.dtype == torch.float32 and tuple(ids.shape) == (200, K)
    ev = ggnn.Evaluator(base, q, gt, K).evaluate_results(ids)
    assert ev.c_k_query > 0.95, repr(ev)
    # the GPU-built graph queried by the oracle gives the same answer as the GPU query
    graph = eng.get_graph(0)
    o_ids, o_d = orc.query(base, q, graph.graph[0].view.numpy(),
                           graph.translation[3].view.numpy().reshape(-1),
                           graph.nn1_stats.view.numpy().reshape(-1), K, 0.64, 400)
    assert np.array_equal(ids.numpy(), o_ids) and np.array_equal(d.numpy(), o_d)
    # graph sanity: valid ids, no self loops in the local part
    g0 = graph.graph[0].view.numpy()
    assert g0.min() >= 0 and g0.max() < N


def test_end_to_end_shards(orc):
    import ggnn_amd as ggnn
    N, D, K = 8000, 64, 10
    base, q = make_int_data(N, D, 51), make_int_data(100, D, 52)
    eng = ggnn.GGNN()
    eng.set_base(base)
    eng.set_shard_size(2000)
    eng.build(24, 0.5, 1)
    ids, d = eng.query(q, K, 0.7, 400)
    gt, _ = eng.bf_query(q, K)
    rec = np.mean([len(set(a) & set(b)) / K for a, b in zip(ids.numpy(), gt.numpy())])
    assert rec > 0.95
    assert (np.diff(d.numpy(), axis=1) >= 0).all()


# ---------------------------------------------------------------------------------------------
# configuration matrix: element type, dimension (chunk configurations), K (registers per lane)
# ---------------------------------------------------------------------------------------------
def _data(dtype, N, D, seed):
    a = np.random.default_rng(seed).integers(0, 256, (N, D))
    return a.astype(np.uint8) if dtype == "u8" else a.astype(np.float32)


_graph_cache = {}


def _mini_graph(orc, dtype, D, K, measure):
    key = (dtype, D, K, measure)
    if key not in _graph_cache:
        N = 1100
        base = _data(dtype, N, D, 100 + D + K)
        cfg, graph, tr, sel, stats = orc.build(base, K, 0.5, 0, measure=measure,
                                               rng=orc.make_rng(N, 11))
        _graph_cache[key] = dict(N=N, D=D, K=K, base=base, cfg=cfg, graph=graph, tr=tr, sel=sel,
                                 stats=stats)
    return _graph_cache[key]


# ("f32", 960, 24, 1) is BASELINE configs[2] (GIST1M shape: cosine AND 960 dimensions together)
CONFIGS = [("u8", 128, 24, 0), ("f32", 96, 24, 0), ("f32", 256, 24, 0), ("f32", 960, 24, 0),
           ("u8", 960, 24, 0), ("f32", 128, 40, 0), ("f32", 64, 60, 0), ("f32", 128, 24, 1),
           ("f32", 32, 8, 0), ("f32", 960, 24, 1)]


@pytest.mark.parametrize("dtype,D,K,measure", CONFIGS)
def test_config_matrix_query_top_merge(ops, orc, dtype, D, K, measure):
    g = _mini_graph(orc, dtype, D, K, measure)
    c = g["cfg"]
    exact = measure == 0
    q = _data(dtype, 48, D, 7)
    graph0 = g["graph"][:g["N"]]
    d_base = dev(g["base"])
    # query
    ids, d = ops.query(d_base, dev(q), dev(graph0), dev(start_points(g)), dev(g["stats"]), 10,
                       0.7, 300, measure)
    # cosine: inexact float arithmetic even on integer data -- the oracle sums in the kernels'
    # order (orc.wave_order), then everything is compared bit for bit as well
    orc.set_wave_order(not exact)
    o_ids, o_d = orc.query(g["base"], q, graph0, start_points(g), g["stats"], 10, 0.7, 300, measure)
    assert np.array_equal(ids.cpu().numpy(), o_ids) and np.array_equal(d.cpu().numpy(), o_d)
    # top (layer 0 and 1)
    for layer in (0, 1):
        tr_l = None if layer == 0 else g["tr"][c.STs_offsets[layer]:c.STs_offsets[layer] + c.Ns[layer]]
        S, S_off = (c.S0, c.S0_off) if layer == 0 else (c.S, 0)
        gr, nn1 = ops.top(d_base, K, None if tr_l is None else dev(tr_l), c.Ns[layer], S, S_off,
                          layer, measure)
        o_gr, o_nn1 = orc.top(g["base"], K, tr_l, c.Ns[layer], S, S_off, layer, measure)
        assert np.array_equal(gr.cpu().numpy(), o_gr)
        assert np.array_equal(nn1.cpu().numpy(), o_nn1)
    # merge 3 -> 0 and 2 -> 1
    for top, btm in ((3, 0), (2, 1)):
        gb, nn1 = ops.merge(d_base, c, dev(g["graph"]), dev(g["tr"]), dev(g["sel"]),
                            dev(g["stats"]), 0.5, top, btm, measure)
        o_gb, o_nn1 = orc.merge(g["base"], c, g["graph"], g["tr"], g["sel"], g["stats"], 0.5, top,
                                btm, measure)
        assert np.array_equal(gb.cpu().numpy(), o_gb)
        if btm == 0:
            assert np.array_equal(nn1.cpu().numpy(), o_nn1)
    orc.set_wave_order(False)


@pytest.mark.parametrize("dtype,D,K,measure", [("u8", 128, 24, 0), ("f32", 960, 24, 0),
                                               ("f32", 64, 60, 0), ("f32", 96, 24, 0),
                                               ("f32", 128, 24, 1), ("u8", 128, 24, 1),
                                               ("f32", 960, 24, 1)])
def test_config_matrix_sym(ops, orc, dtype, D, K, measure):
    g = _mini_graph(orc, dtype, D, K, measure)
    c = g["cfg"]
    KF = K // 2
    Nl = 150
    graph_l = g["graph"][:c.N].copy()
    sb = np.full((c.N, KF), -1, np.int32)
    sa = np.zeros(c.N, np.uint32)
    orc.margin_reset()
    orc.sym(g["base"], K, graph_l, None, g["stats"], 0.5, sb, sa, first_n=0, count=Nl,
            measure=measure)
    if measure == 1:
        # cosine distances are inexact: the oracle repeats the run with the kernels' summation
        # order, then requests and slot counters agree bit for bit
        sb = np.full((c.N, KF), -1, np.int32)
        sa = np.zeros(c.N, np.uint32)
        with orc.wave_order():
            orc.sym(g["base"], K, graph_l, None, g["stats"], 0.5, sb, sa, first_n=0, count=Nl,
                    measure=measure)
        d_sb = torch.full((c.N, KF), -1, dtype=torch.int32, device="cuda")
        d_sa = torch.zeros(c.N, dtype=torch.int32, device="cuda")
        d_base, d_graph, d_stats = dev(g["base"]), dev(graph_l), dev(g["stats"])
        for n in range(Nl):
            ops.sym(d_base, K, d_graph, None, d_stats, 0.5, d_sb, d_sa, measure, first_n=n, count=1)
        assert np.array_equal(d_sb.cpu().numpy(), sb)
        assert np.array_equal(d_sa.cpu().numpy().astype(np.uint32), sa)
        return
    assert orc.margin_min() > 1e-5
    d_sb = torch.full((c.N, KF), -1, dtype=torch.int32, device="cuda")
    d_sa = torch.zeros(c.N, dtype=torch.int32, device="cuda")
    d_base, d_graph, d_stats = dev(g["base"]), dev(graph_l), dev(g["stats"])
    for n in range(Nl):
        ops.sym(d_base, K, d_graph, None, d_stats, 0.5, d_sb, d_sa, measure, first_n=n, count=1)
    assert np.array_equal(d_sa.cpu().numpy().astype(np.uint32), sa)
    assert np.array_equal(d_sb.cpu().numpy(), sb)


def test_unsupported_shapes_fail_loudly(ops):
    base = torch.zeros((100, 30), dtype=torch.float32, device="cuda")   # 120-byte rows
    with pytest.raises(RuntimeError, match="multiple of 16 bytes"):
        ops.bf_query(base, base[:4].contiguous(), 5)


# ---------------------------------------------------------------------------------------------
# bf_query, MFMA path (taken for >= 256 queries, D <= 128, k <= 56): Q x B^T pre-selection on
# the matrix cores + exact direct-form re-rank -> same answers as the scan kernel / the oracle
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,N,D,Nq,K", [("f32", 20000, 128, 300, 10), ("u8", 20000, 128, 257, 10),
                                            ("f32", 9000, 96, 400, 24), ("f32", 5000, 32, 1000, 1),
                                            ("f32", 33333, 100, 256, 50), ("f32", 4100, 4, 260, 5),
                                            ("f32", 30000, 128, 300, 100), ("u8", 9000, 64, 256, 248),
                                            ("f32", 9000, 256, 300, 10), ("f32", 6000, 960, 260, 10),
                                            ("u8", 8000, 960, 256, 20), ("f32", 5000, 132, 256, 10),
                                            ("f32", 4500, 1024, 256, 5)])
def test_bf_mfma_int_exact(ops, orc, dtype, N, D, Nq, K):
    base, q = _data(dtype, N, D, 61), _data(dtype, Nq, D, 62)
    ids, d = ops.bf_query(dev(base), dev(q), K)
    o_ids, o_d = orc.bf_query(base, q, K)
    assert np.array_equal(ids.cpu().numpy(), o_ids)
    assert np.array_equal(d.cpu().numpy(), o_d)


def test_bf_query_baseline_config0_shape(ops, orc):
    """BASELINE configs[0] at its exact shape: 10 000 x 128 float32 base, 10 000 queries, k = 10
    (S-int of SURVEY 8(d): seeds 1234 / 4321).  The whole batch runs on the GPU; 256 evenly spaced
    queries are compared with the oracle (the CPU restatement needs seconds for these, minutes for
    all), and every returned row is checked to be sorted."""
    base, q = make_int_data(10_000, 128, 1234), make_int_data(10_000, 128, 4321)
    ids, d = ops.bf_query(dev(base), dev(q), 10)
    ids, d = ids.cpu().numpy(), d.cpu().numpy()
    assert ids.shape == (10_000, 10) and (np.diff(d, axis=1) >= 0).all()
    pick = np.arange(0, 10_000, 39)[:256]
    o_ids, o_d = orc.bf_query(base, q[pick], 10)
    assert np.array_equal(ids[pick], o_ids)
    assert np.array_equal(d[pick], o_d)


def test_bf_mfma_ties_and_duplicates(ops, orc):
    base = make_int_data(3000, 64, 3)
    base = np.concatenate([base, base, base])           # every distance occurs three times
    q = np.concatenate([make_int_data(200, 64, 4), base[:100]])  # includes exact hits (d = 0)
    ids, d = ops.bf_query(dev(base), dev(q), 12)
    o_ids, o_d = orc.bf_query(base, q, 12)
    assert np.array_equal(ids.cpu().numpy(), o_ids)
    assert np.array_equal(d.cpu().numpy(), o_d)


@pytest.mark.parametrize("measure,D", [(0, 128), (1, 128), (0, 960), (1, 960)])
def test_bf_mfma_float_tolerance(ops, orc, measure, D):
    base, q = make_uni_data(12000, D, 7), make_uni_data(300, D, 8)
    ids, d = ops.bf_query(dev(base), dev(q), 10, measure)
    o_ids, o_d = orc.bf_query(base, q, 10, measure)
    np.testing.assert_allclose(d.cpu().numpy(), o_d, rtol=RTOL, atol=cos_atol(D) if measure else 0)
    assert_topk_parity(base, q, ids.cpu().numpy(), d.cpu().numpy(), o_ids, 10, measure)


# ---------------------------------------------------------------------------------------------
# larger K: four sorted-list registers per lane (KQuery <= 239, KBuild up to the reference's
# effective limit), and arbitrary D through zero-padded rows in the engine
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K,iters", [(120, 512), (200, 512), (239, 1024), (240, 1024), (400, 1024),
                                     (495, 2048), (496, 2048)])
def test_query_large_k_exact(ops, orc, small_graph, K, iters):
    g = small_graph
    q = make_int_data(24, g["D"], 91)
    graph0 = g["graph"][:g["N"]]
    ids, d = ops.query(dev(g["base"]), dev(q), dev(graph0), dev(start_points(g)), dev(g["stats"]),
                       K, 0.8, iters)
    o_ids, o_d = orc.query(g["base"], q, graph0, start_points(g), g["stats"], K, 0.8, iters)
    assert np.array_equal(ids.cpu().numpy(), o_ids) and np.array_equal(d.cpu().numpy(), o_d)


def test_large_kbuild_merge_sym_exact(ops, orc):
    N, D, K = 3000, 32, 120      # merge: sorted 160 (3 regs -> R=4), sym: sorted 96 (R=2)
    base = make_int_data(N, D, 93)
    cfg, graph, tr, sel, stats = orc.build(base, K, 0.5, 0, rng=orc.make_rng(N, 3))
    d_base = dev(base)
    gb, nn1 = ops.merge(d_base, cfg, dev(graph), dev(tr), dev(sel), dev(stats), 0.5, 3, 0)
    o_gb, o_nn1 = orc.merge(base, cfg, graph, tr, sel, stats, 0.5, 3, 0)
    assert np.array_equal(gb.cpu().numpy(), o_gb) and np.array_equal(nn1.cpu().numpy(), o_nn1)
    KF, Nl = K // 2, 60
    graph_l = graph[:N].copy()
    sb = np.full((N, KF), -1, np.int32)
    sa = np.zeros(N, np.uint32)
    orc.margin_reset()
    orc.sym(base, K, graph_l, None, stats, 0.5, sb, sa, first_n=0, count=Nl)
    assert orc.margin_min() > 1e-5
    d_sb = torch.full((N, KF), -1, dtype=torch.int32, device="cuda")
    d_sa = torch.zeros(N, dtype=torch.int32, device="cuda")
    d_graph, d_stats = dev(graph_l), dev(stats)
    for n in range(Nl):
        ops.sym(d_base, K, d_graph, None, d_stats, 0.5, d_sb, d_sa, first_n=n, count=1)
    assert np.array_equal(d_sa.cpu().numpy().astype(np.uint32), sa)
    assert np.array_equal(d_sb.cpu().numpy(), sb)


@pytest.mark.parametrize("dtype,D", [("f32", 30), ("f32", 101), ("u8", 100), ("f32", 1)])
def test_engine_pads_arbitrary_dimensions(orc, dtype, D):
    import ggnn_amd as ggnn
    N, K = 3000, 10
    base, q = _data(dtype, N, D, 95), _data(dtype, 300, D, 96)
    eng = ggnn.GGNN()
    eng.set_base(base)
    gt, gd = eng.bf_query(q, K)
    o_ids, o_d = orc.bf_query(base, q, K)
    assert np.array_equal(gd.numpy(), o_d)
    if D > 1:  # D=1 on 256 integer values is all ties at equal distances: ids still match
        assert np.array_equal(gt.numpy(), o_ids)
    eng.build(24, 0.5, 1)
    ids, d = eng.query(q, K, 0.8, 400)
    graph = eng.get_graph(0)
    assert graph.config["D"] == D
    o = orc.query(base, q, graph.graph[0].view.numpy(), graph.translation[3].view.numpy().reshape(-1),
                  graph.nn1_stats.view.numpy().reshape(-1), K, 0.8, 400)
    assert np.array_equal(ids.numpy(), o[0]) and np.array_equal(d.numpy(), o[1])


@pytest.mark.parametrize("K,iters,ps", [(240, 400, False), (300, 512, False), (1000, 1024, False),
                                        (600, 1000, True), (1007, 2000, False), (1500, 2048, False),
                                        (2031, 2048, True), (2040, 2048, False), (2047, 512, False)])
def test_query_lds_list_for_very_large_k(ops, orc, small_graph, K, iters, ps):
    """KQuery > 239: 8 / 16 / 32 list registers per lane up to KQuery 495 / 1007 / 2031 (the
    pre-screen is not used there), above that the sorted list lives in LDS (wave64 port of the
    literal shift-insert)."""
    g = small_graph
    q = make_int_data(12, g["D"], 97)
    graph0 = g["graph"][:g["N"]]
    b = dev(g["base"])
    ids, d, nd, npop = ops.query(b, dev(q), dev(graph0), dev(start_points(g)),
                                 dev(g["stats"]), K, 0.8, iters, counters=True,
                                 prescreen=ops.prescreen_encode(b) if ps else None)
    o_ids, o_d, o_nd, o_np = orc.query(g["base"], q, graph0, start_points(g), g["stats"], K, 0.8,
                                       iters, counters=True)
    assert np.array_equal(ids.cpu().numpy(), o_ids) and np.array_equal(d.cpu().numpy(), o_d)
    assert np.array_equal(npop.cpu().numpy().astype(np.uint32), o_np)
    assert np.array_equal(nd.cpu().numpy().astype(np.uint32), o_nd)


def test_two_dimensional_grids_beyond_2_20_blocks(ops, orc):
    """More than 2^20 workgroups are launched as a 2-D grid (the AQL packet holds the grid size in
    work-items as 32 bits: a 1-D grid of one wave per point is silently truncated beyond 2^26
    points).  N = 1.1M points makes `top` use blockIdx.y > 0; the result must still equal the
    oracle for EVERY point."""
    N, D, K = 1_100_000, 4, 8
    base = make_int_data(N, D, 111)
    cfg = orc.graph_config(N, D, K)
    graph, nn1 = ops.top(dev(base), K, None, N, cfg.S0, cfg.S0_off, 0)
    o_graph, o_nn1 = orc.top(base, K, None, N, cfg.S0, cfg.S0_off, 0)
    assert np.array_equal(graph.cpu().numpy(), o_graph)
    assert np.array_equal(nn1.cpu().numpy(), o_nn1)
    u = ops.uniform(300_000_000, 7, 0)          # 1.17M blocks of 256 threads
    tail = u[-1_000_000:].cpu().numpy()
    assert tail.min() > 0.0 and tail.max() <= 1.0 and abs(tail.mean() - 0.5) < 0.01


# ---------------------------------------------------------------------------------------------
# exact pre-screen of the float32 / Euclidean query (extension; must not change any result)
# ---------------------------------------------------------------------------------------------
def _quarter_data(N, D, seed):
    """multiples of 1/4 in [0, 256): every fp32 sum is exact, yet the 8-bit coding is lossy"""
    return (np.random.default_rng(seed).integers(0, 1024, (N, D)) / 4.0).astype(np.float32)


def _clustered(N, D, seed, offset=0.0, scale=1.0):
    rng = np.random.default_rng(seed)
    centres = rng.normal(size=(32, D)) * 4.0
    x = centres[rng.integers(0, 32, N)] + rng.normal(size=(N, D))
    return (x * scale + offset).astype(np.float32)


def test_prescreen_sizes_and_lossless_integers(ops):
    base = dev(make_int_data(3000, 128, 5))
    # code rows never straddle more lines than they need: powers of two up to 64, then 64-byte steps
    assert ops.prescreen_sizes(3000, 128)[0] == 128 and ops.prescreen_sizes(3000, 100)[0] == 128
    assert ops.prescreen_sizes(3000, 96)[0] == 128 and ops.prescreen_sizes(3000, 64)[0] == 64
    assert ops.prescreen_sizes(3000, 200)[0] == 256 and ops.prescreen_sizes(3000, 960)[0] == 960
    codes, params = ops.prescreen_encode(base)
    p = params.cpu().numpy()
    assert p[0] == 1.0 and p[1] == 1.0 and p[2] == 0.0 and p[4] == 1.0
    offs = p[8:8 + 128]
    assert np.array_equal(codes.cpu().numpy().astype(np.float32) + offs, base.cpu().numpy())


def test_prescreen_refuses_non_finite(ops):
    base = make_uni_data(1000, 64, 3)
    base[17, 5] = np.inf
    _, params = ops.prescreen_encode(dev(base))
    assert params.cpu().numpy()[4] == 0.0
    base[17, 5] = np.nan
    _, params = ops.prescreen_encode(dev(base))
    assert params.cpu().numpy()[4] == 0.0


@pytest.mark.parametrize("maker,D", [(_quarter_data, 128), (_clustered, 128), (_clustered, 100),
                                     (make_uni_data, 64), (_clustered, 256), (_clustered, 960),
                                     (make_int_data, 2048)])
def test_prescreen_bound_never_exceeds_the_float_distance(ops, orc, maker, D):
    N, Nq, M = 4000, 48, 96
    base, q = maker(N, D, 71), maker(Nq, D, 72)
    Dp = (D + 3) // 4 * 4
    if Dp != D:
        base = np.pad(base, ((0, 0), (0, Dp - D)))
        q = np.pad(q, ((0, 0), (0, Dp - D)))
    codes, params = ops.prescreen_encode(dev(base))
    assert params.cpu().numpy()[4] == 1.0
    rng = np.random.default_rng(73)
    cand = rng.integers(0, N, (Nq, M)).astype(np.int32)
    # The reference pushes iff d < criteria, so the pre-screen must not reject at any criteria
    # above the float32 distance d.  A float32 evaluation (any summation order) lies within
    # (D+8) * 2^-24 of the exact value, so test at a criteria below every possible evaluation.
    d = ((q[:, None, :].astype(np.float64) - base[cand].astype(np.float64)) ** 2).sum(-1)
    d32 = d.astype(np.float32)
    just_above = (d * (1.0 - 3.0 * (Dp + 8) * 2.0 ** -24)).astype(np.float32)
    rej, _ = ops.prescreen_probe(codes, params, dev(q), dev(cand), dev(just_above))
    assert int(rej.sum()) == 0
    rej, _ = ops.prescreen_probe(codes, params, dev(q), dev(cand),
                                 dev(np.full_like(d32, np.inf)))
    assert int(rej.sum()) == 0
    # and it is useful: at half the true distance nearly everything is rejected
    rej, _ = ops.prescreen_probe(codes, params, dev(q), dev(cand), dev(d32 * np.float32(0.5)))
    assert rej.float().mean().item() > 0.9


@pytest.mark.parametrize("maker,kw", [(_quarter_data, {}), (make_int_data, {}), (_clustered, {}),
                                      (_clustered, dict(offset=1000.0)),
                                      (_clustered, dict(scale=1e-3, offset=-5.0)),
                                      (make_uni_data, {})])
@pytest.mark.parametrize("D", [128, 72])
def test_query_prescreened_equals_plain_query(ops, orc, maker, kw, D):
    N, K = 3000, 24
    base, q = maker(N, D, 81, **kw), maker(200, D, 82, **kw)
    cfg, graph, tr, sel, stats = orc.build(base, K, 0.5, 1, rng=orc.make_rng(N, 7))
    start = tr[cfg.STs_offsets[3]:cfg.STs_offsets[3] + cfg.Ns[3]]
    b, qq, g0, st, ss = dev(base), dev(q), dev(graph[:N]), dev(start), dev(stats)
    ps = ops.prescreen_encode(b)
    assert ps[1].cpu().numpy()[4] == 1.0
    for kq, tau, iters in ((10, 0.6, 200), (1, 0.9, 100), (100, 0.5, 400), (300, 0.4, 512)):
        plain = ops.query(b, qq, g0, st, ss, kq, tau, iters, counters=True)
        fast = ops.query(b, qq, g0, st, ss, kq, tau, iters, counters=True, prescreen=ps)
        for x, y in zip(plain, fast):
            assert torch.equal(x, y)
    if maker in (_quarter_data, make_int_data):
        o = orc.query(base, q, graph[:N], start, stats, 10, 0.6, 200, counters=True)
        fast = ops.query(b, qq, g0, st, ss, 10, 0.6, 200, counters=True, prescreen=ps)
        assert np.array_equal(fast[0].cpu().numpy(), o[0])
        assert np.array_equal(fast[1].cpu().numpy(), o[1])
        assert np.array_equal(fast[2].cpu().numpy().astype(np.uint32), o[2])
        assert np.array_equal(fast[3].cpu().numpy().astype(np.uint32), o[3])


@pytest.mark.parametrize("D", [256, 960])
def test_query_prescreened_wide_rows(ops, orc, D):
    N, K = 2000, 24
    base, q = _clustered(N, D, 91), _clustered(64, D, 92)
    cfg, graph, tr, sel, stats = orc.build(base, K, 0.5, 0, rng=orc.make_rng(N, 8))
    start = tr[cfg.STs_offsets[3]:cfg.STs_offsets[3] + cfg.Ns[3]]
    b, qq, g0, st, ss = dev(base), dev(q), dev(graph[:N]), dev(start), dev(stats)
    ps = ops.prescreen_encode(b)
    plain = ops.query(b, qq, g0, st, ss, 10, 0.7, 200, counters=True)
    fast = ops.query(b, qq, g0, st, ss, 10, 0.7, 200, counters=True, prescreen=ps)
    for x, y in zip(plain, fast):
        assert torch.equal(x, y)


@pytest.mark.parametrize("top,btm", [(1, 0), (3, 0), (3, 2), (2, 1)])
def test_merge_prescreened_exact(ops, orc, small_graph, top, btm):
    g = small_graph
    c = g["cfg"]
    b = dev(g["base"])
    ps = ops.prescreen_encode(b)
    gb, nn1, nd = ops.merge(b, c, dev(g["graph"]), dev(g["tr"]), dev(g["sel"]), dev(g["stats"]),
                            0.5, top, btm, counters=True, prescreen=ps)
    o_gb, o_nn1, o_nd = orc.merge(g["base"], c, g["graph"], g["tr"], g["sel"], g["stats"], 0.5,
                                  top, btm, counters=True)
    assert np.array_equal(gb.cpu().numpy(), o_gb)
    assert np.array_equal(nd.cpu().numpy().astype(np.uint32), o_nd)
    if btm == 0:
        assert np.array_equal(nn1.cpu().numpy(), o_nn1)


@pytest.mark.parametrize("maker,D", [(_clustered, 128), (_quarter_data, 128), (_clustered, 200),
                                     (make_uni_data, 64)])
def test_merge_prescreened_equals_plain_merge(ops, orc, maker, D):
    N, K = 2500, 24
    base = maker(N, D, 101)
    cfg, graph, tr, sel, stats = orc.build(base, K, 0.5, 0, rng=orc.make_rng(N, 9))
    b, ga, ta, sa, ss = dev(base), dev(graph), dev(tr), dev(sel), dev(stats)
    ps = ops.prescreen_encode(b)
    for top, btm in ((3, 0), (2, 0), (3, 1)):
        plain = ops.merge(b, cfg, ga, ta, sa, ss, 0.5, top, btm, counters=True)
        fast = ops.merge(b, cfg, ga, ta, sa, ss, 0.5, top, btm, counters=True, prescreen=ps)
        for x, y in zip(plain, fast):
            assert torch.equal(x, y)


# --- cosine -----------------------------------------------------------------------------------
def _cos_dist64(q, x):
    qn = np.sqrt((q.astype(np.float64) ** 2).sum(-1))
    xn = np.sqrt((x.astype(np.float64) ** 2).sum(-1))
    dot = (q.astype(np.float64) * x.astype(np.float64)).sum(-1)
    with np.errstate(invalid="ignore", divide="ignore"):
        d = np.abs(1.0 - dot / (qn * xn))
    return np.where(qn * xn > 0, d, 1.0)


@pytest.mark.parametrize("maker,D", [(_clustered, 128), (make_int_data, 128), (_clustered, 960),
                                     (make_uni_data, 64)])
def test_prescreen_cosine_bound_never_exceeds_the_float_distance(ops, maker, D):
    N, Nq, M = 4000, 48, 96
    base, q = maker(N, D, 71), maker(Nq, D, 72)
    base[5] = 0.0  # a zero row has distance 1 to everything
    codes, params = ops.prescreen_encode(dev(base), 1)
    p = params.cpu().numpy()
    assert p[4] == 1.0 and p[7] == 1.0
    rng = np.random.default_rng(73)
    cand = rng.integers(0, N, (Nq, M)).astype(np.int32)
    cand[:, 0] = 5
    d = _cos_dist64(q[:, None, :], base[cand])
    # a float32 evaluation lies within (D+8) * 2^-24 (absolute) of the exact value
    below = (d - 3.0 * (D + 8) * 2.0 ** -24).astype(np.float32)
    rej, _ = ops.prescreen_probe(codes, params, dev(q), dev(cand), dev(below), 1)
    assert int(rej.sum()) == 0
    rej, _ = ops.prescreen_probe(codes, params, dev(q), dev(cand), dev((d * 0.5).astype(np.float32)),
                                 1)
    assert rej.float().mean().item() > 0.9


@pytest.mark.parametrize("maker,D", [(_clustered, 128), (make_int_data, 128), (_clustered, 320),
                                     (_clustered, 960)])
def test_query_and_merge_prescreened_cosine_equal_plain(ops, orc, maker, D):
    N, K = 2500, 24
    base, q = maker(N, D, 111), maker(150, D, 112)
    base[7] = 0.0
    q[3] = 0.0  # zero-norm query: the reference returns distance 1 for every candidate
    cfg, graph, tr, sel, stats = orc.build(base, K, 0.5, 0, measure=1, rng=orc.make_rng(N, 10))
    start = tr[cfg.STs_offsets[3]:cfg.STs_offsets[3] + cfg.Ns[3]]
    b, qq, ga, ta, sa, ss = dev(base), dev(q), dev(graph), dev(tr), dev(sel), dev(stats)
    g0, st = dev(graph[:N]), dev(start)
    ps = ops.prescreen_encode(b, 1)
    assert ps[1].cpu().numpy()[4] == 1.0
    for kq, tau, iters in ((10, 0.6, 200), (100, 0.5, 400)):
        plain = ops.query(b, qq, g0, st, ss, kq, tau, iters, 1, counters=True)
        fast = ops.query(b, qq, g0, st, ss, kq, tau, iters, 1, counters=True, prescreen=ps)
        for x, y in zip(plain, fast):
            assert torch.equal(x, y)
    for top, btm in ((3, 0), (2, 1)):
        plain = ops.merge(b, cfg, ga, ta, sa, ss, 0.5, top, btm, 1, counters=True)
        fast = ops.merge(b, cfg, ga, ta, sa, ss, 0.5, top, btm, 1, counters=True, prescreen=ps)
        for x, y in zip(plain, fast):
            assert torch.equal(x, y)


@pytest.mark.parametrize("maker,kw,D,measure", [(make_int_data, {}, 128, 0), (_clustered, {}, 128, 0),
                                                 (_clustered, dict(offset=1000.0), 128, 0),
                                                 (_clustered, {}, 200, 0), (_clustered, {}, 128, 1),
                                                 (_clustered, {}, 960, 1)])
def test_sym_prescreened_equals_plain_sym(ops, orc, maker, kw, D, measure):
    """the sym kernel with the exact pre-screen on the query-point distance: launched one point at
    a time (sym is racy by design) it must leave sym_buffer / sym_atomic exactly as the plain
    kernel does, on a layer-0 graph and on an upper layer (translation)"""
    N, K = 2500, 24
    KF = K // 2
    base = maker(N, D, 131, **kw)
    cfg, graph, tr, sel, stats = orc.build(base, K, 0.5, 0, measure=measure, rng=orc.make_rng(N, 12))
    b, ss = dev(base), dev(stats)
    ps = ops.prescreen_encode(b, measure)
    assert ps[1].cpu().numpy()[4] == 1.0
    for layer, Nl in ((0, 160), (1, 80)):
        g_l = dev(graph[cfg.Ns_offsets[layer]:cfg.Ns_offsets[layer] + cfg.Ns[layer]].copy())
        t_l = None if layer == 0 else dev(tr[cfg.STs_offsets[layer]:cfg.STs_offsets[layer] + cfg.Ns[layer]])
        out = []
        for pre in (None, ps):
            sb = torch.full((cfg.Ns[layer], KF), -1, dtype=torch.int32, device="cuda")
            sa = torch.zeros(cfg.Ns[layer], dtype=torch.int32, device="cuda")
            for n in range(Nl):
                ops.sym(b, K, g_l, t_l, ss, 0.5, sb, sa, measure, first_n=n, count=1, prescreen=pre)
            out.append((sb, sa))
        assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
        assert int(out[0][1].sum()) > 0     # some inverse links were requested at all


@pytest.mark.parametrize("K", [20, 40])
def test_prescreen_other_graph_degrees(ops, orc, K):
    """KBuild = 40 needs two fetch blocks per pop, KBuild = 20 leaves lanes of a block empty"""
    N, D = 3000, 128
    base, q = _clustered(N, D, 121), _clustered(120, D, 122)
    cfg, graph, tr, sel, stats = orc.build(base, K, 0.5, 0, rng=orc.make_rng(N, 11))
    start = tr[cfg.STs_offsets[3]:cfg.STs_offsets[3] + cfg.Ns[3]]
    b, qq, ga, ta, sa, ss = dev(base), dev(q), dev(graph), dev(tr), dev(sel), dev(stats)
    ps = ops.prescreen_encode(b)
    plain = ops.query(b, qq, dev(graph[:N]), dev(start), ss, 10, 0.7, 200, counters=True)
    fast = ops.query(b, qq, dev(graph[:N]), dev(start), ss, 10, 0.7, 200, counters=True,
                     prescreen=ps)
    for x, y in zip(plain, fast):
        assert torch.equal(x, y)
    plain = ops.merge(b, cfg, ga, ta, sa, ss, 0.5, 3, 0, counters=True)
    fast = ops.merge(b, cfg, ga, ta, sa, ss, 0.5, 3, 0, counters=True, prescreen=ps)
    for x, y in zip(plain, fast):
        assert torch.equal(x, y)
