"""GPU tests of the ggnn.GGNN surface beyond the kernels: device-resident inputs, results on the
GPU, shards, store/load, counters (all through the C-ABI handle API)."""
import os

import numpy as np
import pytest

from conftest import make_int_data

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def recall(a, b):
    return np.mean([len(set(x) & set(y)) / len(y) for x, y in zip(a, b)])


@pytest.fixture(scope="module")
def data():
    return make_int_data(5000, 64, 71), make_int_data(150, 64, 72)


def test_gpu_inputs_and_results_on_gpu(data):
    import ggnn_amd as ggnn
    base, q = data
    eng = ggnn.GGNN()
    eng.set_base(torch.from_numpy(base).cuda())           # examples/python/ggnn_pytorch_gpu_data.py
    eng.set_return_results_on_gpu(True)
    eng.build(24, 0.5)
    ids, d = eng.query(torch.from_numpy(q).cuda(), 10, 0.7, 400)
    assert ids.is_cuda and d.is_cuda and tuple(ids.shape) == (150, 10)
    gt, gd = eng.bf_query(torch.from_numpy(q).cuda(), 10)
    assert gt.is_cuda
    assert recall(ids.cpu().numpy(), gt.cpu().numpy()) > 0.95
    # the same engine answers host queries identically
    eng.set_return_results_on_gpu(False)
    ids2, d2 = eng.query(q, 10, 0.7, 400)
    assert not ids2.is_cuda
    assert np.array_equal(ids2.numpy(), ids.cpu().numpy()) and np.array_equal(d2.numpy(), d.cpu().numpy())


def test_base_reference_is_borrowed(data):
    import ggnn_amd as ggnn
    base, q = data
    t = torch.from_numpy(base).cuda()
    eng = ggnn.GGNN()
    eng.set_base_reference(t)
    eng.build(24, 0.5, 1)
    ids, _ = eng.query(q, 10, 0.7)
    gt, _ = eng.bf_query(q, 10)
    assert recall(ids.numpy(), gt.numpy()) > 0.95


def test_shards_results_on_gpu_are_unmerged_sorted_rows(data, orc):
    import ggnn_amd as ggnn
    base, q = data
    eng = ggnn.GGNN()
    eng.set_base(base)
    eng.set_shard_size(1250)                                # 4 shards on one GPU
    eng.set_return_results_on_gpu(True)
    eng.build(24, 0.5, 1)
    ids, d = eng.query(q, 10, 0.7)
    assert tuple(ids.shape) == (150, 40)                    # ggnn.cuh:108-113: N*K sorted results
    dn = d.cpu().numpy()
    assert (np.diff(dn, axis=1) >= 0).all()
    assert ids.cpu().numpy().min() >= 0 and ids.cpu().numpy().max() < 5000
    eng.set_return_results_on_gpu(False)
    ids2, d2 = eng.query(q, 10, 0.7)
    assert np.array_equal(ids2.numpy(), ids.cpu().numpy()[:, :10])
    # every shard is an independent graph: per-shard oracle query + reference merge == engine
    for s in range(4):
        g = eng.get_graph(s)
        assert g.config["N"] == 1250
    parts_i, parts_d = [], []
    for s in range(4):
        g = eng.get_graph(s)
        o = orc.query(base[s * 1250:(s + 1) * 1250], q, g.graph[0].view.numpy(),
                      g.translation[3].view.numpy().reshape(-1),
                      g.nn1_stats.view.numpy().reshape(-1), 10, 0.7, 400)
        parts_i.append(o[0] + s * 1250)
        parts_d.append(o[1])
    allr = orc.sort_shard_results(np.concatenate(parts_i, 1), np.concatenate(parts_d, 1))
    assert np.array_equal(allr[1], dn)
    uniq = np.ones_like(dn, bool)
    uniq[:, 1:] &= dn[:, 1:] != dn[:, :-1]
    uniq[:, :-1] &= dn[:, :-1] != dn[:, 1:]
    assert np.array_equal(allr[0][uniq], ids.cpu().numpy()[uniq])


def test_store_load_roundtrip(data, tmp_path):
    import ggnn_amd as ggnn
    base, q = data
    a = ggnn.GGNN()
    a.set_base(base)
    a.set_working_directory(str(tmp_path))
    a.set_shard_size(2500)
    a.build(24, 0.5, 1)
    a.store()
    cfg = a.get_graph(0).config
    expect = (cfg["N_all"] * 24 + 2 * cfg["ST_all"]) * 4 + 8   # graph.cpp:48-91 pool layout
    for s in (0, 1):
        assert os.path.getsize(tmp_path / f"part_{s}.ggnn") == expect
    ids_a, d_a = a.query(q, 10, 0.7)
    b = ggnn.GGNN()
    b.set_base(base)
    b.set_working_directory(str(tmp_path))
    b.set_shard_size(2500)
    b.load(24)
    ids_b, d_b = b.query(q, 10, 0.7)
    assert np.array_equal(ids_a.numpy(), ids_b.numpy()) and np.array_equal(d_a.numpy(), d_b.numpy())
    c = ggnn.GGNN()
    c.set_base(base)
    c.set_working_directory(str(tmp_path / "empty"))
    with pytest.raises(RuntimeError, match="graph file"):
        c.load(24)


def test_counters_and_timing(data):
    import ggnn_amd as ggnn
    base, q = data
    eng = ggnn.GGNN()
    eng.set_base(base)
    eng.build(24, 0.5, 0)
    eng.set_collect_counters(True)
    eng.query(q, 10, 0.6, 200)
    c = eng.last_query_counters()
    t = eng.last_timing_ms()
    assert c["n_pop"] > 0 and c["n_dist"] >= 32 * 150 and c["n_pop"] <= 200 * 150
    assert t["build_ms"] > 0 and t["query_ms"] > 0


def test_misuse_after_build(data):
    import ggnn_amd as ggnn
    base, q = data
    eng = ggnn.GGNN()
    eng.set_base(base)
    eng.build(24, 0.5, 0)
    with pytest.raises(RuntimeError, match="already been built"):
        eng.build(24, 0.5, 0)
    with pytest.raises(RuntimeError, match="cannot be changed"):
        eng.set_base(base)
    with pytest.raises(RuntimeError, match="data type"):
        eng.query(q.astype(np.uint8), 10, 0.5)
    with pytest.raises(RuntimeError, match="dimension"):
        eng.query(np.zeros((3, 32), np.float32), 10, 0.5)
    with pytest.raises(RuntimeError, match="does not exist"):
        eng.get_graph(3)
    e2 = ggnn.GGNN()
    e2.set_base(base)
    e2.set_shard_size(1234)
    with pytest.raises(RuntimeError, match="evenly divisible"):
        e2.build(24, 0.5)


def test_edge_cases_empty_and_tiny(orc):
    import ggnn_amd as ggnn
    base = make_int_data(3000, 32, 81)
    eng = ggnn.GGNN()
    eng.set_base(base)
    eng.build(24, 0.5, 0)
    ids, d = eng.query(np.zeros((0, 32), np.float32), 10, 0.5)       # empty query set
    assert tuple(ids.shape) == (0, 10) and tuple(d.shape) == (0, 10)
    ids, d = eng.bf_query(np.zeros((0, 32), np.float32), 5)
    assert tuple(ids.shape) == (0, 5)
    q1 = make_int_data(1, 32, 82)                                     # single query, k = 1
    ids, d = eng.query(q1, 1, 0.5)
    o = orc.bf_query(base, q1, 1)
    assert tuple(ids.shape) == (1, 1) and d[0, 0] >= o[1][0, 0]
    # N=100, K=8 -> G=2, S0=12 but 16 points per segment would have to be promoted
    tiny = ggnn.GGNN()
    tiny.set_base(make_int_data(100, 32, 83))
    with pytest.raises(RuntimeError, match="too small"):
        tiny.build(8, 0.5)
    # N=100, K=24 -> G=1: a degenerate but valid hierarchy, as in the reference
    ok = ggnn.GGNN()
    ok.set_base(make_int_data(100, 32, 83))
    ok.build(24, 0.5)
    ids, d = ok.query(q1, 5, 0.9)
    assert ids.numpy().min() >= 0 and ids.numpy().max() < 100
    too_small = ggnn.GGNN()
    too_small.set_base(make_int_data(20, 32, 83))
    with pytest.raises(RuntimeError, match="too small"):
        too_small.build(24, 0.5)


@pytest.mark.parametrize("K", [2, 3, 8])
def test_minimum_kbuild_build_and_query(orc, K):
    import ggnn_amd as ggnn
    base, q = make_int_data(4000, 32, 84), make_int_data(64, 32, 85)
    eng = ggnn.GGNN()
    eng.set_base(base)
    eng.build(K, 0.5, 1)
    g = eng.get_graph(0)
    g0 = g.graph[0].view.numpy()
    assert g0.shape == (4000, K) and g0.min() >= 0 and g0.max() < 4000
    ids, d = eng.query(q, 5, 0.9, 400)
    o = orc.query(base, q, g0, g.translation[3].view.numpy().reshape(-1),
                  g.nn1_stats.view.numpy().reshape(-1), 5, 0.9, 400)
    assert np.array_equal(ids.numpy(), o[0]) and np.array_equal(d.numpy(), o[1])


def test_duplicate_points_zero_distances(orc):
    """every vector appears four times: zero distances, ties everywhere (merge's nn1 skips zero
    distances, merge_layer.cu:147-157)"""
    import ggnn_amd as ggnn
    u = make_int_data(1000, 32, 86)
    base = np.concatenate([u, u, u, u])
    eng = ggnn.GGNN()
    eng.set_base(base)
    eng.build(24, 0.5, 1)
    g = eng.get_graph(0)
    stats = g.nn1_stats.view.numpy().reshape(-1)
    assert np.isfinite(stats).all() and stats[1] >= stats[0] >= 0
    q = u[:50].copy()
    ids, d = eng.query(q, 10, 0.8, 400)
    gt, gd = eng.bf_query(q, 10)
    o_gt, o_gd = orc.bf_query(base, q, 10)
    assert np.array_equal(gt.numpy(), o_gt) and np.array_equal(gd.numpy(), o_gd)
    assert (gd.numpy()[:, :4] == 0).all()
    o = orc.query(base, q, g.graph[0].view.numpy(), g.translation[3].view.numpy().reshape(-1),
                  stats, 10, 0.8, 400)
    assert np.array_equal(ids.numpy(), o[0]) and np.array_equal(d.numpy(), o[1])


def test_rccl_exchange_single_rank_world(orc, monkeypatch):
    """GGNN_EXCHANGE=rccl routes even a one-GPU handle through the RCCL exchange of the multi-GPU
    path (ncclCommInitAll over the handle's devices, grouped ncclAllGather of ids and distances,
    slice merge on the device, slice copies to the host): a one-rank world on this one-GPU box, the
    same code the 8-GPU handle runs.  Results must equal the plain path's."""
    import ggnn_amd as ggnn
    N, D, K = 8000, 64, 10
    base, q = make_int_data(N, D, 187), make_int_data(333, D, 188)
    eng = ggnn.GGNN()
    eng.set_base(base)
    eng.set_shard_size(2000)       # 4 resident shards: rows of K * 4 sorted candidates
    eng.build(24, 0.5, 1)
    ids, d = eng.query(q, K, 0.7, 200)
    assert eng.last_exchange() == "none"
    monkeypatch.setenv("GGNN_EXCHANGE", "rccl")
    ids2, d2 = eng.query(q, K, 0.7, 200)
    assert eng.last_exchange() == "rccl"
    assert torch.equal(ids, ids2) and torch.equal(d, d2)
    ids3, d3 = eng.query(q[:1], K, 0.7, 200)   # fewer queries than ranks * anything: slice math
    assert torch.equal(ids[:1], ids3) and torch.equal(d[:1], d3)


def test_overlapped_shard_launches_equal_sequential():
    """several resident shards: the per-shard kernels run concurrently on a few streams; with the
    work counters switched on they run one at a time -- same sorted rows either way"""
    import ggnn_amd as ggnn
    base, q = make_int_data(12000, 64, 195), make_int_data(500, 64, 196)
    eng = ggnn.GGNN()
    eng.set_base(base)
    eng.set_shard_size(2000)                 # 6 shards on 4 streams
    eng.set_return_results_on_gpu(True)
    eng.build(24, 0.5, 1)
    ids, d = eng.query(q, 10, 0.7, 200)
    assert tuple(ids.shape) == (500, 60)
    eng.set_collect_counters(True)
    ids2, d2 = eng.query(q, 10, 0.7, 200)
    assert eng.last_query_counters()["n_pop"] > 0
    eng.set_collect_counters(False)
    ids3, d3 = eng.query(q, 10, 0.7, 200)
    assert torch.equal(ids, ids2) and torch.equal(d, d2)
    assert torch.equal(ids, ids3) and torch.equal(d, d3)
    assert (d[:, 1:] >= d[:, :-1]).all()     # rows sorted across shards


@pytest.mark.parametrize("shard_size", [0, 3000])
def test_query_async_equals_blocking_query(shard_size):
    """query_async / synchronize (serving extension): batches enqueued on alternating slots give
    the rows of the blocking results-on-GPU call, for one and for several resident shards"""
    import ggnn_amd as ggnn
    base, q = make_int_data(12000, 64, 197), make_int_data(700, 64, 198)
    eng = ggnn.GGNN()
    eng.set_base(base)
    if shard_size:
        eng.set_shard_size(shard_size)
    eng.set_return_results_on_gpu(True)
    eng.build(24, 0.5, 1)
    qd = torch.from_numpy(q).cuda()
    qd2 = torch.from_numpy(q[::-1].copy()).cuda()
    ref = eng.query(qd, 10, 0.7, 200)
    ref2 = eng.query(qd2, 10, 0.7, 200)
    outs = []
    for i in range(6):
        outs.append(eng.query_async(qd if i % 2 == 0 else qd2, 10, 0.7, 200, slot=i))
    eng.synchronize()
    for i, (ids, d) in enumerate(outs):
        want = ref if i % 2 == 0 else ref2
        assert torch.equal(ids, want[0]) and torch.equal(d, want[1])
    with pytest.raises(RuntimeError, match="on the GPU"):
        eng.query_async(q, 10, 0.7, 200)


def test_failed_load_rolls_back(tmp_path):
    """a load() that fails half way must not leave a handle that claims a graph."""
    import ggnn_amd as ggnn
    base, q = make_int_data(4000, 32, 191), make_int_data(16, 32, 192)
    eng = ggnn.GGNN()
    eng.set_base(base)
    eng.set_working_directory(str(tmp_path))
    eng.set_shard_size(2000)
    eng.build(24, 0.5, 1)
    eng.store()
    ref_ids, ref_d = eng.query(q, 10, 0.6, 200)
    os.remove(os.path.join(str(tmp_path), "part_1.ggnn"))
    eng2 = ggnn.GGNN()
    eng2.set_base(base)
    eng2.set_working_directory(str(tmp_path))
    eng2.set_shard_size(2000)
    with pytest.raises(RuntimeError, match="missing or mismatching graph file"):
        eng2.load(24)
    with pytest.raises(RuntimeError, match="no graph to query"):
        eng2.query(q, 10, 0.6, 200)
    with pytest.raises(RuntimeError, match="No graph has been built"):
        eng2.get_graph(0)
    eng2.set_base(base)            # allowed again: the handle is back in its set_base state
    eng2.set_shard_size(2000)
    eng2.build(24, 0.5, 1)         # and a build succeeds
    ids, d = eng2.query(q, 10, 0.6, 200)
    assert tuple(ids.shape) == (16, 10)


@pytest.mark.parametrize("shard_size", [4000, 2000])
def test_in_process_multi_gpu(orc, shard_size):
    """set_gpus([...]) with several entries drives several device contexts from one handle (host
    thread per GPU, candidates copied to the first GPU, device merge) like the reference's
    ggnn_main_multi_gpu example.  The test box has one GPU, so both contexts use device 0."""
    import ggnn_amd as ggnn
    N, D, K = 8000, 64, 10
    base, q = make_int_data(N, D, 87), make_int_data(120, D, 88)
    eng = ggnn.GGNN()
    eng.set_base(base)
    eng.set_gpus([0, 0])
    with pytest.raises(RuntimeError, match="evenly divisible"):
        eng.build(24, 0.5, 1)      # as in the reference, several GPUs need an explicit shard size
    eng.set_shard_size(shard_size)
    eng.build(24, 0.5, 1)
    ids, d = eng.query(q, K, 0.7, 400)
    assert tuple(ids.shape) == (120, K) and not ids.is_cuda
    n_shard = shard_size
    spg = N // n_shard // 2
    # reference semantics: per-GPU sorted rows, then ResultMerger with offset g*spg*N_shard
    parts_i, parts_d = [], []
    for gpu in range(2):
        rows_i, rows_d = [], []
        for s in range(spg):
            gs = gpu * spg + s
            g = eng.get_graph(gs)
            lo = gs * n_shard
            o = orc.query(base[lo:lo + n_shard], q, g.graph[0].view.numpy(),
                          g.translation[3].view.numpy().reshape(-1),
                          g.nn1_stats.view.numpy().reshape(-1), K, 0.7, 400)
            rows_i.append(o[0] + s * n_shard)
            rows_d.append(o[1])
        si, sd = orc.sort_shard_results(np.concatenate(rows_i, 1), np.concatenate(rows_d, 1))
        parts_i.append(si)
        parts_d.append(sd)
    r_ids, r_d = orc.merge_results(parts_i, parts_d, K, spg, n_shard)
    assert eng.last_exchange() == "copy"   # both contexts share device 0: RCCL needs distinct GPUs
    assert np.array_equal(d.numpy(), r_d)
    uniq = np.ones_like(r_d, bool)
    uniq[:, 1:] &= r_d[:, 1:] != r_d[:, :-1]
    uniq[:, :-1] &= r_d[:, :-1] != r_d[:, 1:]
    assert np.array_equal(ids.numpy()[uniq], r_ids[uniq])
    gt, _ = orc.bf_query(base, q, K)
    assert recall(ids.numpy(), gt) > 0.9
    # reference restrictions for several GPUs (ggnn.cu:299-301, 338-339)
    with pytest.raises(RuntimeError, match="single GPU"):
        eng.bf_query(q, K)
    eng.set_return_results_on_gpu(True)
    with pytest.raises(RuntimeError, match="single GPU"):
        eng.query(q, K, 0.7)


@pytest.mark.parametrize("nq", [37, 5, 64])
def test_eight_contexts_gather_and_slice_merges(orc, nq):
    """What an 8-GPU handle does behind its all-gather, on the one-GPU box: set_gpus([0] * 8) with
    two shards per context, hook EXCHANGE = 3 ("gather": every context gathers all rows by peer
    copies, merges ITS 1/8 slice of the queries and returns it through its own staging buffer --
    the RCCL path's structure without RCCL, which refuses two ranks on one device).  Query counts
    that 8 does not divide (37: slices of 5,5,5,5,5,5,5,2) and fewer queries than contexts (5:
    three empty slices), unsplit and as two half-batches in flight -- against the reference's
    per-GPU sort + ResultMerger (gpu_instance.cu:745-790, result_merger.cpp:51-149)."""
    import ggnn_amd as ggnn
    from ggnn_amd import _lib
    N, D, K, NSH = 8000, 64, 10, 500
    base, q = make_int_data(N, D, 187), make_int_data(nq, D, 188)
    eng = ggnn.GGNN()
    eng.set_base(base)
    eng.set_gpus([0] * 8)
    eng.set_shard_size(NSH)
    eng.build(24, 0.5, 1)
    spg = N // NSH // 8
    parts_i, parts_d = [], []
    for gpu in range(8):
        rows_i, rows_d = [], []
        for s in range(spg):
            gs = gpu * spg + s
            g = eng.get_graph(gs)
            lo = gs * NSH
            o = orc.query(base[lo:lo + NSH], q, g.graph[0].view.numpy(),
                          g.translation[3].view.numpy().reshape(-1),
                          g.nn1_stats.view.numpy().reshape(-1), K, 0.7, 200)
            rows_i.append(o[0] + s * NSH)
            rows_d.append(o[1])
        si, sd = orc.sort_shard_results(np.concatenate(rows_i, 1), np.concatenate(rows_d, 1))
        parts_i.append(si)
        parts_d.append(sd)
    r_ids, r_d = orc.merge_results(parts_i, parts_d, K, spg, NSH)
    uniq = np.ones_like(r_d, bool)
    uniq[:, 1:] &= r_d[:, 1:] != r_d[:, :-1]
    uniq[:, :-1] &= r_d[:, :-1] != r_d[:, 1:]
    uniq[:, -1] = False      # (a tie of the K-th with the first row left out: either id is right)
    for exchange, split in ((3, 0), (3, 1), (2, 0)):
        with _lib.hooks(EXCHANGE=exchange, QUERY_SPLIT=split):
            ids, d = eng.query(q, K, 0.7, 200)
        assert eng.last_exchange() == ("gather" if exchange == 3 else "copy")
        assert eng.last_query_parts() == (2 if split else 1)
        assert np.array_equal(d.numpy(), r_d), (exchange, split)
        assert np.array_equal(ids.numpy()[uniq], r_ids[uniq]), (exchange, split)


def test_uint8_and_cosine_through_the_api(orc):
    """uint8 base (SIFT-like) and cosine measure through the public surface incl. the Evaluator"""
    import ggnn_amd as ggnn
    r = np.random.default_rng(91)
    base = r.integers(0, 256, (6000, 128)).astype(np.uint8)
    q = r.integers(0, 256, (100, 128)).astype(np.uint8)
    for measure in (ggnn.DistanceMeasure.Euclidean, ggnn.DistanceMeasure.Cosine):
        eng = ggnn.GGNN()
        eng.set_base(ggnn.UCharDataset(base))
        eng.build(24, 0.5, 1, measure)
        ids, d = eng.query(ggnn.UCharDataset(q), 10, 0.8, 400, measure)
        gt, gd = eng.bf_query(q, 10, measure)
        o_gt, o_gd = orc.bf_query(base, q, 10, int(measure))
        if measure == ggnn.DistanceMeasure.Euclidean:
            assert np.array_equal(gt.numpy(), o_gt) and np.array_equal(gd.numpy(), o_gd)
        else:
            np.testing.assert_allclose(gd.numpy(), o_gd, rtol=1e-4, atol=1e-7)
        ev = ggnn.Evaluator(base, q, gt, 10, measure).evaluate_results(ids)
        assert ev.c_k_query > 0.9, repr(ev)


def test_prescreen_toggle_gives_identical_results():
    # one graph (construction is not run-to-run deterministic: sym uses atomics, as in the
    # reference), queried with the pre-screen on, off and on again
    import ggnn_amd as ggnn
    rng = np.random.default_rng(5)
    centres = rng.normal(size=(16, 128)) * 3
    base = (centres[rng.integers(0, 16, 6000)] + rng.normal(size=(6000, 128))).astype(np.float32)
    q = (centres[rng.integers(0, 16, 300)] + rng.normal(size=(300, 128))).astype(np.float32)
    g = ggnn.GGNN()
    g.set_collect_counters(True)
    g.set_base(base)
    g.build(24, 0.5, 1)
    out = []
    for enable in (True, False, True):
        g.set_prescreen(enable)
        ids, d = g.query(q, 10, 0.8, 200)
        out.append((np.asarray(ids), np.asarray(d), g.last_query_counters()))
    for o in out[1:]:
        assert np.array_equal(out[0][0], o[0])
        assert np.array_equal(out[0][1], o[1])
        assert out[0][2] == o[2]


def test_prescreen_follows_the_measure():
    # one handle, Euclidean build then cosine and Euclidean queries: the pre-screen copy is
    # re-coded for the measure in use and results equal the plain kernels
    import ggnn_amd as ggnn
    rng = np.random.default_rng(6)
    centres = rng.normal(size=(16, 128)) * 3
    base = (centres[rng.integers(0, 16, 5000)] + rng.normal(size=(5000, 128))).astype(np.float32)
    q = (centres[rng.integers(0, 16, 200)] + rng.normal(size=(200, 128))).astype(np.float32)
    g = ggnn.GGNN()
    g.set_base(base)
    g.build(24, 0.5, 1)
    res = {}
    for measure in (ggnn.DistanceMeasure.Cosine, ggnn.DistanceMeasure.Euclidean,
                    ggnn.DistanceMeasure.Cosine):
        for enable in (True, False):
            g.set_prescreen(enable)
            ids, d = g.query(q, 10, 0.7, 200, measure)
            res.setdefault((measure, enable), (np.asarray(ids), np.asarray(d)))
            assert np.array_equal(res[(measure, enable)][0], np.asarray(ids))
        assert np.array_equal(res[(measure, True)][0], res[(measure, False)][0])
        assert np.array_equal(res[(measure, True)][1], res[(measure, False)][1])


@pytest.mark.parametrize("D,measure", [(100, 0), (100, 1), (67, 0)])
def test_prescreen_with_padded_rows(D, measure):
    # rows that the engine pads to a multiple of 16 bytes (and the codes to 16 dimensions)
    import ggnn_amd as ggnn
    rng = np.random.default_rng(8)
    centres = rng.normal(size=(16, D)) * 3
    base = (centres[rng.integers(0, 16, 5000)] + rng.normal(size=(5000, D))).astype(np.float32)
    q = (centres[rng.integers(0, 16, 150)] + rng.normal(size=(150, D))).astype(np.float32)
    base[11] = 0.0
    g = ggnn.GGNN()
    g.set_collect_counters(True)
    g.set_base(base)
    g.build(24, 0.5, 1, measure)
    res = []
    for enable in (True, False):
        g.set_prescreen(enable)
        ids, d = g.query(q, 10, 0.8, 200, measure)
        res.append((np.asarray(ids), np.asarray(d), g.last_query_counters()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert res[0][2] == res[1][2]
    rows = g.last_query_rows_read()
    assert rows["code_rows"] == 0 and rows["float_rows"] == res[1][2]["n_dist"]


@pytest.mark.parametrize("where", ["device", "pinned", "pageable"])
def test_query_async_multi_gpu_handle_copy_exchange(where):
    """query_async on a handle that drives several GPU contexts (both on device 0 here, so the
    exchange runs through peer copies + one merge): merged [Nq, K] results equal the blocking
    query(), for a query in device memory, page-locked and pageable host memory"""
    import ggnn_amd as ggnn
    N, D, K = 8000, 64, 10
    base, q = make_int_data(N, D, 287), make_int_data(250, D, 288)
    q2 = q[::-1].copy()
    eng = ggnn.GGNN()
    eng.set_base(base)
    eng.set_gpus([0, 0])
    eng.set_shard_size(2000)          # two contexts x two resident shards
    eng.build(24, 0.5, 1)
    ref, ref2 = eng.query(q, K, 0.7, 200), eng.query(q2, K, 0.7, 200)
    assert eng.last_exchange() == "copy"

    def put(x):
        t = torch.from_numpy(x)
        return t.cuda() if where == "device" else (t.pin_memory() if where == "pinned" else t)

    tickets = [eng.query_async(put(q if i % 2 == 0 else q2), K, 0.7, 200, slot=i) for i in range(5)]
    eng.synchronize()
    for i, t in enumerate(tickets):
        want = ref if i % 2 == 0 else ref2
        assert t.done and tuple(t.ids.shape) == (250, K)
        assert torch.equal(t.ids.cpu(), want[0]) and torch.equal(t.dists.cpu(), want[1])
    # one slot only
    t = eng.query_async(put(q), K, 0.7, 200, slot=2)
    eng.synchronize(2)
    assert t.done and torch.equal(t.ids.cpu(), ref[0])


def test_query_async_rccl_single_rank_world(monkeypatch):
    """the RCCL form of the asynchronous multi-GPU path on a one-rank world: GGNN_EXCHANGE=rccl
    sends a one-GPU, four-shard handle through communicator creation, the packed grouped
    all-gather on the slot's stream, the slice merge and the asynchronous result copies"""
    import ggnn_amd as ggnn
    N, D, K = 8000, 64, 10
    base, q = make_int_data(N, D, 387), make_int_data(333, D, 388)
    eng = ggnn.GGNN()
    eng.set_base(base)
    eng.set_gpus([0])
    eng.set_shard_size(2000)
    eng.build(24, 0.5, 1)
    ref = eng.query(q, K, 0.7, 200)
    monkeypatch.setenv("GGNN_EXCHANGE", "rccl")
    again = eng.query(q, K, 0.7, 200)
    assert eng.last_exchange() == "rccl"
    assert torch.equal(ref[0], again[0]) and torch.equal(ref[1], again[1])
    qd = torch.from_numpy(q).cuda()
    tickets = [eng.query_async(qd, K, 0.7, 200, slot=i) for i in range(3)]
    tickets.append(eng.query_async(torch.from_numpy(q), K, 0.7, 200, slot=3))   # pinned host query
    eng.synchronize()
    for t in tickets:
        assert tuple(t.ids.shape) == (333, K)
        assert torch.equal(t.ids.cpu(), ref[0]) and torch.equal(t.dists.cpu(), ref[1])
    assert eng.last_exchange() == "rccl"


def test_query_async_ticket_keeps_inputs_alive():
    """the engine streams are invisible to torch's caching allocator.  A temporary
    query tensor whose last Python reference dies right after query_async() must not be recycled
    (and overwritten) while the kernel still reads it: the engine keeps the ticket until
    synchronize()."""
    import gc
    import ggnn_amd as ggnn
    base, q = make_int_data(20000, 64, 411), make_int_data(4000, 64, 412)
    eng = ggnn.GGNN()
    eng.set_base(base)
    eng.set_return_results_on_gpu(True)
    eng.build(24, 0.5, 1)
    qd = torch.from_numpy(q).cuda()
    ref = eng.query(qd, 10, 0.9, 400)
    outs = []
    for i in range(8):
        tmp = qd.clone()                       # a temporary: freed as soon as we drop it
        outs.append(eng.query_async(tmp, 10, 0.9, 400, slot=i % 2))
        ptr = tmp.data_ptr()
        del tmp
        gc.collect()
        # the allocator would hand the same block out again if nothing referenced it
        junk = torch.full((4000, 64), 255.0, device="cuda")
        assert junk.data_ptr() != ptr or i == -1
        del junk
    assert sum(len(v) for v in eng._inflight.values()) == 8
    eng.synchronize()
    assert not eng._inflight
    for ids, d in outs:
        assert torch.equal(ids, ref[0]) and torch.equal(d, ref[1])


def test_query_async_multi_gpu_many_batches_in_flight():
    """Batches of one lane reuse that lane's buffers on every GPU context.  With the copy exchange
    the FIRST GPU copies the other contexts' rows on its own stream, so their owners must not
    start the lane's next local search before that copy has run (found by the full-size dry run
    of bench.py --gpus 2: results of pipelined batches differed from the blocking calls)."""
    import ggnn_amd as ggnn
    from bench import synthetic
    dev_ = torch.device("cuda", 0)
    N, D, K = 400_000, 128, 10
    base = synthetic("lowrank16", N, D, 11, dev_)
    qa = synthetic("lowrank16", 6000, D, 12, dev_)
    qb = synthetic("lowrank16", 6000, D, 13, dev_)
    eng = ggnn.GGNN()
    eng.set_base_reference(base)
    eng.set_gpus([0, 0])
    eng.set_shard_size(100_000)
    eng.build(24, 0.5, 1)
    ra, rb = eng.query(qa, K, 0.9, 175), eng.query(qb, K, 0.9, 175)
    for rounds in range(3):
        tickets = [eng.query_async(qa if i % 2 == 0 else qb, K, 0.9, 175, slot=i % 2)
                   for i in range(12)]
        eng.synchronize()
        for i, t in enumerate(tickets):
            want = ra if i % 2 == 0 else rb
            assert torch.equal(t.ids.cpu(), want[0]) and torch.equal(t.dists.cpu(), want[1]), (rounds, i)


def test_query_async_growing_batches_on_one_slot():
    """A larger batch than any before re-allocates the lane's buffers while smaller batches of the
    same lane are in flight; the peer copies / RCCL kernels of ANOTHER GPU may still use the old
    allocations, which the owning device's hipFree does not wait for -- the engine drains the
    lane on every GPU before it grows a buffer.  (Two contexts on one device here: the logic runs,
    the cross-device window itself needs a second GPU.)"""
    import ggnn_amd as ggnn
    from bench import synthetic
    dev_ = torch.device("cuda", 0)
    N, D, K = 200_000, 128, 10
    base = synthetic("lowrank16", N, D, 21, dev_)
    q = synthetic("lowrank16", 8000, D, 22, dev_)
    eng = ggnn.GGNN()
    eng.set_base_reference(base)
    eng.set_gpus([0, 0])
    eng.set_shard_size(50_000)
    eng.build(24, 0.5, 1)
    sizes = [500, 1000, 3000, 8000, 2000]
    want = [eng.query(q[:n].contiguous(), K, 0.9, 175) for n in sizes]
    tickets = [eng.query_async(q[:n].contiguous(), K, 0.9, 175, slot=0) for n in sizes]
    eng.synchronize()
    for n, t, w in zip(sizes, tickets, want):
        assert torch.equal(t.ids.cpu(), w[0]) and torch.equal(t.dists.cpu(), w[1]), n


@pytest.mark.gpu
def test_blocking_multi_gpu_query_is_split_in_two_half_batches():
    """Several GPUs: a blocking query() runs as two half-batches in flight (the exchange and merge
    of the first overlap the search of the second) -- on by default from 4096 queries, forced here
    from 2 (hook QUERY_SPLIT).  Results are bit-identical to the unsplit call, on the copy path
    (two contexts on device 0) and on the RCCL path (one-rank world), odd query counts included."""
    import ggnn_amd as ggnn
    from ggnn_amd import _lib
    N, D, K = 8000, 64, 10
    base, q = make_int_data(N, D, 487), make_int_data(777, D, 488)
    for gpus, exchange, how in (([0, 0], 0, "copy"), ([0], 1, "rccl")):
        eng = ggnn.GGNN()
        eng.set_base(base)
        eng.set_gpus(gpus)
        eng.set_shard_size(2000)
        eng.build(24, 0.5, 1)
        with _lib.hooks(EXCHANGE=exchange, QUERY_SPLIT=0):
            ref = eng.query(q, K, 0.7, 200)
            assert eng.last_query_parts() == 1 and eng.last_exchange() == how
        with _lib.hooks(EXCHANGE=exchange, QUERY_SPLIT=1):
            for nq in (777, 2, 3, 64):
                ids, d = eng.query(q[:nq], K, 0.7, 200)
                assert eng.last_query_parts() == 2 and eng.last_exchange() == how
                assert torch.equal(ids, ref[0][:nq]) and torch.equal(d, ref[1][:nq]), (how, nq)
            assert eng.last_timing_ms()["query_ms"] > 0
        # default: small batches are not split, large ones are
        with _lib.hooks(EXCHANGE=exchange):
            eng.query(q, K, 0.7, 200)
            assert eng.last_query_parts() == 1
            big = np.concatenate([q] * 6)[:4200]
            ids, d = eng.query(big, K, 0.7, 200)
            assert eng.last_query_parts() == 2
            assert torch.equal(ids[:777], ref[0]) and torch.equal(d[777:1554], ref[1])
        del eng


@pytest.mark.gpu
def test_rccl_failure_falls_back_to_peer_copies():
    """Fault injection (hook RCCL_FAIL_AFTER): the n-th exchange reports an RCCL failure.  The engine
    drains its streams, drops the communicators for good and serves that call and every later one
    through peer copies -- same results, blocking and asynchronous."""
    import ggnn_amd as ggnn
    from ggnn_amd import _lib
    N, D, K = 8000, 64, 10
    base, q = make_int_data(N, D, 587), make_int_data(333, D, 588)
    eng = ggnn.GGNN()
    eng.set_base(base)
    eng.set_gpus([0])
    eng.set_shard_size(2000)
    eng.build(24, 0.5, 1)
    ref = eng.query(q, K, 0.7, 200)
    with _lib.hooks(EXCHANGE=1, QUERY_SPLIT=0):
        a = eng.query(q, K, 0.7, 200)
        assert eng.last_exchange() == "rccl"
        qd = torch.from_numpy(q).cuda()
        t0 = eng.query_async(qd, K, 0.7, 200, slot=1)     # in flight on the communicators
        with _lib.hooks(RCCL_FAIL_AFTER=1):                # the next exchange fails inside RCCL
            b = eng.query(q, K, 0.7, 200)
        assert eng.last_exchange() == "copy"
        eng.synchronize()
        c = eng.query(q, K, 0.7, 200)                      # communicators are gone for good
        assert eng.last_exchange() == "copy"
        t1 = eng.query_async(qd, K, 0.7, 200, slot=2)
        eng.synchronize()
    for ids, d in (a, b, c, (t0.ids.cpu(), t0.dists.cpu()), (t1.ids.cpu(), t1.dists.cpu())):
        assert torch.equal(ids, ref[0]) and torch.equal(d, ref[1])


def test_query_async_tickets_are_bounded_without_synchronize():
    """A serving loop that never calls synchronize() (it waits some other way): the tensors the
    engine holds for its kernels' sake stay bounded -- a slot is drained and released once it
    holds _MAX_TICKETS_PER_SLOT batches."""
    import ggnn_amd as ggnn
    from ggnn_amd import api
    base, q = make_int_data(4000, 64, 687), make_int_data(64, 64, 688)
    eng = ggnn.GGNN()
    eng.set_base(base)
    eng.set_return_results_on_gpu(True)
    eng.build(24, 0.5, 1)
    qd = torch.from_numpy(q).cuda()
    ref = eng.query(qd, 10, 0.7, 100)
    tickets = [eng.query_async(qd, 10, 0.7, 100, slot=1) for _ in range(api._MAX_TICKETS_PER_SLOT + 5)]
    assert len(eng._inflight[1]) == 5
    assert all(t.done for t in tickets[:api._MAX_TICKETS_PER_SLOT])
    eng.synchronize()
    assert all(torch.equal(t.ids, ref[0]) and torch.equal(t.dists, ref[1]) for t in tickets)
