"""BASELINE.json configurations at their defining sizes / combinations (the kernel parity tests
run on graphs of a few thousand points):
  * configs[1]  SIFT1M shape, 1M x 128 f32: build, recall@10 >= 0.99 against the certified
    exact bf_query, pre-screen on/off bit-identical, and the CPU oracle's traversal on a sample
    of queries over the GPU-built graph -- bit-identical ids, distances and counters;
  * configs[2]  GIST1M shape, 1M x 960 f32 with the cosine measure: build, exact ground truth
    checked in float64, pre-screen on/off bit-identical, oracle traversal on a sample;
  * configs[4]  SIFT1B/8-style shard whose rows span more than 2^32 bytes (uint8): 64-bit row
    addressing in build, query and bf_query, ids close to the top of the range come back.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def recall_at_k(ids, gt):
    return (ids.unsqueeze(2) == gt.unsqueeze(1)).any(2).float().mean().item()


# fixed points of THIS suite for the recall assertions (not bench.OPERATING_POINTS): measured
# 0.9938-0.9942 (1M x 128, tau 0.9 / 200 iterations) and 0.9939 (12.5M x 96, 1.0 / 300)
TEST_POINT_1M = (0.9, 200)
TEST_POINT_12M = (1.0, 300)
BENCH_POINT_FLOOR = 0.9885


def test_sift1m_shape_full_size(orc):
    import ggnn_amd as ggnn
    from bench import synthetic
    dev = torch.device("cuda", 0)
    N, D, Nq, K = 1_000_000, 128, 10_000, 10
    base = synthetic("lowrank16", N, D, 1234, dev)
    query = synthetic("lowrank16", Nq, D, 4321, dev)
    eng = ggnn.GGNN()
    eng.set_base_reference(base)
    eng.set_return_results_on_gpu(True)
    eng.build(24, 0.5, 2)
    gt, gt_d = eng.bf_query(query, K)
    assert eng.last_bf_query_rescanned() < Nq // 100
    eng.set_collect_counters(True)
    # Recall is asserted at a FIXED, slightly conservative point of this test (round-5 advisor
    # finding: asserting >= 0.99 at the benchmark's own tuned point, 0.9913-0.9917, ties the suite
    # to a margin of 0.001 that a legitimate kernel or build change may move), on the tuning query
    # set and on one the points were never tuned on
    held = synthetic("lowrank16", Nq, D, 8642, dev)
    gt_held = eng.bf_query(held, K)[0]
    assert recall_at_k(eng.query(query, K, *TEST_POINT_1M)[0], gt) >= 0.99
    assert recall_at_k(eng.query(held, K, *TEST_POINT_1M)[0], gt_held) >= 0.99
    # the operating point bench.py quotes `value` at (its table, not a copy of it) selects the
    # KERNEL the parity checks below run through; its recall is the bench line's to report, here
    # only a floor under the build-to-build spread
    from bench import OPERATING_POINTS
    tau, iters = OPERATING_POINTS[(1, N, D, "f32", "l2")][:2]
    ids, d = eng.query(query, K, tau, iters)
    cnt = eng.last_query_counters()
    rows = eng.last_query_rows_read()
    assert recall_at_k(ids, gt) >= BENCH_POINT_FLOOR
    assert recall_at_k(eng.query(held, K, tau, iters)[0], gt_held) >= BENCH_POINT_FLOOR
    eng.query(query, K, tau, iters)
    assert rows["code_rows"] > 0 and rows["float_rows"] < cnt["n_dist"] // 3   # pre-screen active
    eng.set_prescreen(False)
    ids2, d2 = eng.query(query, K, tau, iters)
    assert eng.last_query_counters() == cnt
    assert torch.equal(ids, ids2) and torch.equal(d, d2)
    eng.set_prescreen(True)
    # ground truth spot check against float64 on the host (integer-valued data: exact)
    b64 = base[:, :].cpu().numpy()
    for n in (0, 17, 4242):
        dd = ((b64.astype(np.float64) - query[n].cpu().numpy().astype(np.float64)) ** 2).sum(1)
        order = np.lexsort((np.arange(N), dd))[:K]
        assert np.array_equal(order, gt[n].cpu().numpy()) and np.array_equal(dd[order], gt_d[n].cpu().numpy())
    # the oracle's traversal on the GPU-built graph (200 queries): bit-identical
    g = eng.get_graph(0)
    graph0 = g.graph[0].view.numpy()
    start = g.translation[3].view.numpy().reshape(-1)
    stats = g.nn1_stats.view.numpy().reshape(-1)
    q_h = query[:200].cpu().numpy()
    o_ids, o_d, o_nd, o_np = orc.query(b64, q_h, graph0, start, stats, K, tau, iters, counters=True)
    assert np.array_equal(ids[:200].cpu().numpy(), o_ids)
    assert np.array_equal(d[:200].cpu().numpy(), o_d)
    eng.query(query[:200].contiguous(), K, tau, iters)
    c200 = eng.last_query_counters()
    assert c200["n_dist"] == int(o_nd.sum()) and c200["n_pop"] == int(o_np.sum())


def test_gist1m_shape_cosine_960(orc):
    """cosine + D = 960 together (the <float, 64, 4, ..., kCos> instantiations), through the API"""
    import ggnn_amd as ggnn
    from bench import synthetic
    dev = torch.device("cuda", 0)
    N, D, Nq, K = 1_000_000, 960, 1000, 10
    # (the 16-dimensional latent: the 32-dimensional one needs tau 2.0 / 1000 iterations for 0.97
    # at this size, see profiles/r02_bench_n1.json "datasets" for how recall moves with it)
    base = synthetic("lowrank16", N, D, 1234, dev)
    query = synthetic("lowrank16", Nq, D, 4321, dev)
    eng = ggnn.GGNN()
    eng.set_base_reference(base)
    eng.set_return_results_on_gpu(True)
    eng.build(24, 0.5, 2, ggnn.DistanceMeasure.Cosine)
    gt, gt_d = eng.bf_query(query, K, ggnn.DistanceMeasure.Cosine)
    eng.set_collect_counters(True)
    ids, d = eng.query(query, K, 1.0, 400, ggnn.DistanceMeasure.Cosine)
    cnt = eng.last_query_counters()
    assert eng.last_query_rows_read()["code_rows"] > 0
    assert recall_at_k(ids, gt) >= 0.99
    eng.set_prescreen(False)
    ids2, d2 = eng.query(query, K, 1.0, 400, ggnn.DistanceMeasure.Cosine)
    assert eng.last_query_counters() == cnt and torch.equal(ids, ids2) and torch.equal(d, d2)
    # exact ground truth in float64 for a few queries (ties aside, cosine is inexact in float32)
    b64 = base.double()
    bn = b64.norm(dim=1)
    for n in (0, 99, 500):
        q64 = query[n].double()
        dd = (1.0 - (b64 @ q64) / (bn * q64.norm())).abs()
        best = torch.sort(dd).values[:K]
        assert float(dd[gt[n].long()].max()) <= float(best[K - 1]) + 1e-6
        np.testing.assert_allclose(gt_d[n].cpu().numpy(), best.cpu().numpy(), atol=2e-6)
    del b64, bn
    # oracle traversal on the GPU-built graph with the kernels' float summation order: every
    # decision of the cosine search is then the same decision on both sides -- bit for bit
    g = eng.get_graph(0)
    with orc.wave_order():
        o_ids, o_d = orc.query(base.cpu().numpy(), query[:64].cpu().numpy(), g.graph[0].view.numpy(),
                               g.translation[3].view.numpy().reshape(-1),
                               g.nn1_stats.view.numpy().reshape(-1), K, 1.0, 400, 1)
    assert np.array_equal(ids[:64].cpu().numpy(), o_ids)
    assert np.array_equal(d[:64].cpu().numpy(), o_d)


def test_uint8_shard_beyond_4gib():
    """36M x 128 uint8 = 4.6 GB of rows (> 2^32 bytes): the C5 addressing at a size that builds
    in well under a minute.  Queries that ARE base rows from the last 2^32-crossing part of the
    shard must find themselves (distance 0, their own id) through the graph and through bf."""
    import ggnn_amd as ggnn
    from bench import synthetic
    dev = torch.device("cuda", 0)
    N, D, K = 36_000_000, 128, 10
    assert N * D > 2 ** 32
    base = torch.empty((N, D), dtype=torch.uint8, device=dev)
    for lo in range(0, N, 4_000_000):
        hi = min(N, lo + 4_000_000)
        base[lo:hi] = synthetic("lowrank16", hi - lo, D, 1234 + lo, dev).to(torch.uint8)
    probe = torch.arange(N - 2000, N, 4, device=dev)            # ids beyond byte offset 2^32
    assert int(probe.min()) * D > 2 ** 32
    query = torch.cat([base[probe], synthetic("lowrank16", 524, D, 4321, dev).to(torch.uint8)])
    eng = ggnn.GGNN()
    eng.set_base_reference(base)
    eng.set_return_results_on_gpu(True)
    eng.build(24, 0.5, 2)
    gt, gt_d = eng.bf_query(query, K)
    assert int(gt.max()) > N - 2000 and int(gt.min()) >= 0
    assert torch.equal(gt[:500, 0], probe.int()) or (gt_d[:500, 0] == 0).all()
    assert (gt_d[:500, 0] == 0).all()
    ids, d = eng.query(query, K, 1.5, 400)
    assert recall_at_k(ids, gt) >= 0.97
    hit = (ids[:500] == probe.int().unsqueeze(1)).any(1).float().mean().item()
    assert hit >= 0.98, hit          # the rows at the top of the range are found by the traversal
    # exact check of a few bf answers on the host, rows fetched by 64-bit index
    for n in (0, 250, 499, 600):
        cand = gt[n].long()
        dd = ((base[cand].float() - query[n].float()) ** 2).sum(1)
        assert torch.equal(dd, gt_d[n])


def _mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def _shard_case(orc, N, D, dtype, expect, tau, iters, nq=10_000, n_oracle=100, check_prescreen=False,
                recall_point=None):
    """One BASELINE shard at its defining size: layout equal to SURVEY 8(a) row L, build, recall
    against the certified exact bf_query, oracle traversal over the GPU-built graph."""
    import ggnn_amd as ggnn
    from bench import synthetic
    dev = torch.device("cuda", 0)
    K = 10
    base = torch.empty((N, D), dtype=dtype, device=dev)
    for lo in range(0, N, 5_000_000):
        hi = min(N, lo + 5_000_000)
        base[lo:hi] = synthetic("lowrank16", hi - lo, D, 1234 + lo, dev).to(dtype)
    query = synthetic("lowrank16", nq, D, 4321, dev).to(dtype)
    eng = ggnn.GGNN()
    eng.set_base_reference(base)
    eng.set_return_results_on_gpu(True)
    eng.build(24, 0.5, 2)
    gt, gt_d = eng.bf_query(query, K)
    assert eng.last_bf_query_rescanned() <= nq // 100
    # recall_point: the suite's own fixed point for the recall assertion when (tau, iters) is the
    # benchmark's tuned point (see TEST_POINT_1M above), which then only has to hold a floor
    if recall_point is not None:
        rec = recall_at_k(eng.query(query, K, *recall_point)[0], gt)
        assert rec >= 0.99, rec
    eng.set_collect_counters(True)
    ids, d = eng.query(query, K, tau, iters)
    cnt = eng.last_query_counters()
    rec = recall_at_k(ids, gt)
    assert rec >= (BENCH_POINT_FLOOR if recall_point is not None else 0.99), rec
    if check_prescreen:
        rows = eng.last_query_rows_read()
        assert rows["code_rows"] > 0
        eng.set_prescreen(False)
        ids2, d2 = eng.query(query, K, tau, iters)
        assert eng.last_query_counters() == cnt
        assert torch.equal(ids, ids2) and torch.equal(d, d2)
        eng.set_prescreen(True)
    # exact ground truth spot check with 64-bit row indices
    for n in (0, nq // 2, nq - 1):
        cand = gt[n].long()
        dd = ((base[cand].float() - query[n].float()) ** 2).sum(1)
        assert torch.equal(dd, gt_d[n])
    # layout: SURVEY 8(a) row L (values the survey obtained from the reference's graph_config.cpp)
    view = _graph_view(eng)
    cfg = view.config.as_dict()
    for key, val in expect.items():
        assert cfg[key] == val, (key, cfg[key], val)
    # oracle traversal (needs the rows and layer 0 of the graph on the host)
    need_gb = (N * D * base.element_size() + N * 24 * 4) / 1e9
    if _mem_available_gb() < 2.5 * need_gb + 16:
        pytest.skip(f"host memory too small for the oracle part ({need_gb:.0f} GB of rows + graph)")
    base_h = base.cpu().numpy()
    graph0 = torch.empty((N, 24), dtype=torch.int32)
    _copy_d2h(graph0, view.graph, N * 24 * 4)
    ST = cfg["STs_offsets"][3]
    start = torch.empty(cfg["Ns"][3], dtype=torch.int32)
    _copy_d2h(start, view.translation + ST * 4, cfg["Ns"][3] * 4)
    stats = torch.empty(2, dtype=torch.float32)
    _copy_d2h(stats, view.nn1_stats, 8)
    q_h = query[:n_oracle].cpu().numpy()
    o_ids, o_d, o_nd, o_np = orc.query(base_h, q_h, graph0.numpy(), start.numpy(), stats.numpy(),
                                       K, tau, iters, counters=True)
    assert np.array_equal(ids[:n_oracle].cpu().numpy(), o_ids)
    assert np.array_equal(d[:n_oracle].cpu().numpy(), o_d)
    eng.query(query[:n_oracle].contiguous(), K, tau, iters)
    c = eng.last_query_counters()
    assert c["n_dist"] == int(o_nd.sum()) and c["n_pop"] == int(o_np.sum())


def _graph_view(eng):
    import ctypes as C
    from ggnn_amd import _lib
    view = _lib.GraphView()
    _lib.check(_lib.lib().ggnn_get_graph(eng._h, 0, C.byref(view)), eng._h)
    return view


def _copy_d2h(dst, src_ptr, nbytes):
    """device pointer of the engine's graph pool -> pinned-free host tensor, without the full
    get_graph() copy (12 GB for the C5 shard)"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    assert hip.hipMemcpy(dst.data_ptr(), src_ptr, nbytes, 2) == 0   # hipMemcpyDeviceToHost


def test_deep100m_shard_full_size(orc):
    """configs[3]: one of the 8 shards of DEEP100M, 12.5M x 96 f32.  G = 73 with SG = 0 and
    SG_off = 32: only the first 32 of each 73 lower segments promote a point (graph_config.cpp
    :94-97), a selection layout no smaller test reaches."""
    from bench import OPERATING_POINTS
    tau, iters = OPERATING_POINTS[(1, 12_500_000, 96, "f32", "l2")][:2]   # the bench's own point
    _shard_case(orc, 12_500_000, 96, torch.float32,
                dict(G=73, S=32, S0=32, S0_off=51_456, SG=0, SG_off=32, N_all=12_672_896),
                tau, iters, check_prescreen=True, recall_point=TEST_POINT_12M)


def test_sift1b_shard_full_size(orc):
    """configs[4]: one of the 8 shards of SIFT1B, 125M x 128 uint8 (16 GB of rows, 12 GB of
    graph): G = 157, SG = 0, every index path beyond 2^32 bytes."""
    _shard_case(orc, 125_000_000, 128, torch.uint8,
                dict(G=157, S=32, S0=32, S0_off=1_163_424, SG=0, SG_off=32, N_all=125_793_824),
                1.5, 400)
