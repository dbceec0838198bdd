"""CPU tests of the oracle: golden GraphConfig values, hand-derived known-answer tests written
from the cited reference lines, and internal consistency (no GPU)."""
import json
import os

import numpy as np
import pytest

from conftest import make_int_data, make_uni_data

HERE = os.path.dirname(os.path.abspath(__file__))


def test_graph_config_golden(orc):
    """outputs of the reference's graph_config.cpp."""
    with open(os.path.join(HERE, "golden", "graph_config.json")) as f:
        golden = json.load(f)
    for g in golden:
        c = orc.graph_config(g["N"], g["D"], g["KBuild"]).as_dict()
        for k, v in g.items():
            assert c[k] == v, (g["N"], k, c[k], v)


def test_graph_config_invariants(orc):
    for N in (1000, 4096, 10_000, 123_457, 1_000_000):
        for K in (8, 24, 40, 96):
            c = orc.graph_config(N, 64, K).as_dict()
            assert c["G"] ** 3 * c["S0"] + c["S0_off"] == N
            assert c["SG"] * c["G"] + c["SG_off"] == c["S"]
            assert c["S"] % 32 == 0 and c["S"] > K // 2
            assert c["Ns"] == [N, c["G"] ** 2 * c["S"], c["G"] * c["S"], c["S"]]
            assert c["N_all"] == sum(c["Ns"]) and c["ST_all"] == sum(c["Ns"][1:])


def test_query_sizing_appendix_b(orc):
    """SURVEY.md Appendix B worked values (query_kernels.cu:55-110)."""
    for (K, it, D), (cache, sorted_, block) in {
            (10, 200, 128): (256, 64, 32), (10, 400, 128): (512, 32, 32),
            (100, 2000, 128): (2048, 128, 128), (10, 400, 960): (512, 32, 256)}.items():
        s = orc.query_sizing(D, K, it)
        assert (s.cache_size, s.sorted_size, s.block_dim_x) == (cache, sorted_, block)


def test_sizing_helpers_pinned_to_reference_def_h(orc):
    """oracle/_ref/libggnn_ref_def.so is the reference's own include/ggnn/base/def.h compiled
    unchanged with g++ (oracle/ref_def_harness.cpp): bit_ceil / next_multiple<32> / align8 of the
    oracle -- and through test_cabi.py::test_query_sizing_matches_oracle of the engine -- must
    agree with it on every value the sizing rules can see (query_kernels.cu:55-110: KQuery <= 6000,
    max_iterations <= 8192, D <= 4096)."""
    import ctypes as C
    if not os.path.exists(orc.REF_DEF_SO):
        pytest.skip("oracle/_ref was not built (reference tree not mounted at build time)")
    ref = C.CDLL(orc.REF_DEF_SO)
    ref.ref_bit_ceil.restype = C.c_uint32
    ref.ref_next_multiple32.restype = C.c_uint32
    ref.ref_align8.restype = C.c_size_t
    ref.ref_align8.argtypes = [C.c_size_t]
    values = list(range(0, 8300)) + [2 ** k + o for k in range(13, 32) for o in (-1, 0, 1)
                                     if 0 <= 2 ** k + o < 2 ** 31 + 1]
    for v in values:
        assert orc.bit_ceil(v) == ref.ref_bit_ceil(C.c_uint32(v)), v
        assert orc.next_multiple32(v) == ref.ref_next_multiple32(C.c_uint32(v)), v
    for v in list(range(0, 100)) + [2 ** 32 - 1, 2 ** 32, 2 ** 32 + 1, 125_793_824 * 24 * 4 + 13]:
        assert orc.align8(v) == ref.ref_align8(v), v
    assert (ref.ref_measure_euclidean(), ref.ref_measure_cosine()) == (orc.EUCLIDEAN, orc.COSINE)


# ---- KBestList / cache known-answer tests ----------------------------------------------------
def test_kat_q1_ring_wrap_loss(orc):
    """SURVEY.md Q1, derived from simple_knn_cache.cuh:167-172: BEST=2, SORTED=6, head=4,
    prioQ phys[2..5]=(30,-,10,20); push 15 => (30,30,10,15): 20 lost, 30 duplicated."""
    P, POP, XI = orc.op_push, orc.op_pop, orc.op_xi
    ops = [XI(1e9), P(101, 1.0), P(102, 2.0), POP(), POP(), P(10, 10.0), P(20, 20.0),
           P(30, 30.0)]
    k, d, pops, h = orc.cache_script(2, 6, 16, 32, 0.0, ops)
    assert list(k[2:6]) == [30, -1, 10, 20] and h[0] == 4
    k, d, pops, h = orc.cache_script(2, 6, 16, 32, 0.0, ops + [P(15, 15.0)])
    assert list(k[2:6]) == [30, 30, 10, 15]
    assert list(d[2:6]) == [30.0, 30.0, 10.0, 15.0]
    # with the head at BEST (no wrap) the same pushes give the expected sorted queue
    ops2 = [XI(1e9), P(101, 1.0), P(102, 2.0), P(10, 10.0), P(20, 20.0), P(30, 30.0),
            P(15, 15.0)]
    k, d, _, h = orc.cache_script(2, 6, 16, 32, 0.0, ops2)
    assert h[0] == 2 and list(k[:2]) == [101, 102]
    assert list(k[2:6]) == [101, 102, 10, 15]  # queue: 1,2,10,15 (20 and 30 fell off the end)


def test_kat_q2_tie_order(orc):
    """Q2 (simple_knn_cache.cuh:178,207): a new key with an equal distance goes BEFORE existing
    ones in the cache; KBestList keeps insertion order (k_best_list.cuh:92-103)."""
    P = orc.op_push
    k, d, _, _ = orc.cache_script(3, 8, 16, 32, 0.0, [P(1, 5.0), P(2, 5.0), P(3, 5.0)])
    assert list(k[:3]) == [3, 2, 1] and list(k[3:6]) == [3, 2, 1]
    base = np.zeros((4, 4), np.float32)
    base[3] = 1
    ids, dist = orc.bf_query(base, np.zeros((1, 4), np.float32), 3)
    assert list(ids[0]) == [0, 1, 2] and list(dist[0]) == [0, 0, 0]


def test_kat_push_dedup_and_pop_criteria(orc):
    P, POP, XI = orc.op_push, orc.op_pop, orc.op_xi
    # duplicates in best+prioQ are ignored (simple_knn_cache.cuh:131-146)
    k, d, _, _ = orc.cache_script(2, 6, 16, 32, 0.0, [P(7, 3.0), P(7, 1.0), P(8, 2.0)])
    assert list(k[:2]) == [8, 7] and list(d[:2]) == [2.0, 3.0]
    # pop returns EMPTY when dist >= best[BEST-1] + xi (:223)
    ops = [XI(0.5), P(1, 1.0), P(2, 2.0), P(3, 2.4), POP(), POP(), POP(), POP()]
    k, d, pops, h = orc.cache_script(2, 6, 16, 32, 0.0, ops)
    assert list(pops[4:]) == [1, 2, 3, -1]   # 2.4 < 2.0+0.5 is popped, then the queue is empty
    ops = [XI(0.3), P(1, 1.0), P(2, 2.0), P(3, 2.4), POP(), POP(), POP()]
    _, _, pops, _ = orc.cache_script(2, 6, 16, 32, 0.0, ops)
    assert list(pops[4:]) == [1, 2, -1]      # 2.4 >= 2.0+0.3
    assert list(k[6:9]) == [1, 2, 3]         # visited ring


def test_push_is_block_size_independent(orc):
    r = np.random.default_rng(0)
    P, POP, XI, TR = orc.op_push, orc.op_pop, orc.op_xi, orc.op_transform
    for trial in range(60):
        BEST, SORTED = int(r.integers(1, 40)), int(r.choice([64, 96, 128]))
        ops = [XI(1e9)]
        for _ in range(int(r.integers(20, 300))):
            x = r.random()
            ops.append(P(int(r.integers(0, 150)), float(r.integers(0, 40))) if x < 0.6 else
                       POP() if x < 0.97 else TR())
        ref = orc.cache_script(BEST, SORTED, 256, 32, 0.0, ops)
        for block in (64, 128, 512):
            got = orc.cache_script(BEST, SORTED, 256, block, 0.0, ops)
            assert all(np.array_equal(a, b) for a, b in zip(ref, got))


# ---- distances and brute force ---------------------------------------------------------------
def test_distance_formulas(orc):
    base = make_uni_data(10, 100, 1)
    q = make_uni_data(1, 100, 2)
    for i in range(10):
        d = orc.distance(base, q, i, orc.EUCLIDEAN, 32, 4)
        np.testing.assert_allclose(d, ((base[i] - q[0]) ** 2).sum(), rtol=1e-5)
        c = orc.distance(base, q, i, orc.COSINE, 32, 4)
        ref = abs(1 - (base[i] @ q[0]) / np.sqrt((base[i] @ base[i]) * (q[0] @ q[0])))
        np.testing.assert_allclose(c, ref, rtol=1e-4, atol=1e-6)
    z = np.zeros((1, 100), np.float32)
    assert orc.distance(z, q, 0, orc.COSINE, 32, 4) == 1.0  # norm product <= 0 (distance.cuh:157)


def test_bf_query_matches_numpy(orc):
    base, q = make_int_data(3000, 128, 3), make_int_data(50, 128, 4)
    ids, d = orc.bf_query(base, q, 10)
    dm = ((q[:, None, :] - base[None]) ** 2).sum(-1)
    ref = np.argsort(dm, 1, kind="stable")[:, :10]
    assert np.array_equal(ids, ref) and np.array_equal(d, np.take_along_axis(dm, ref, 1))
    bu, qu = base.astype(np.uint8), q.astype(np.uint8)
    ids8, d8 = orc.bf_query(bu, qu, 10)
    assert np.array_equal(ids8, ids) and np.array_equal(d8, d)


# ---- result handling ---------------------------------------------------------------------------
def test_merge_results_semantics(orc):
    """result_merger.cpp:51-149: global id = partition*shards_per_gpu*N_shard + key."""
    a_i = np.array([[5, 6, 7]], np.int32)
    a_d = np.array([[1., 4., 9.]], np.float32)
    b_i = np.array([[1, 2, 3]], np.int32)
    b_d = np.array([[2., 3., 10.]], np.float32)
    ids, d = orc.merge_results([a_i, b_i], [a_d, b_d], 3, 2, 100)
    assert list(d[0]) == [1., 2., 3.] and list(ids[0]) == [5, 201, 202]
    ids, d = orc.merge_results([a_i], [a_d], 3, 1, 100)  # pass-through
    assert list(ids[0]) == [5, 6, 7]


def test_evaluator_counts(orc):
    gt = np.array([[0, 1, 2, 3], [4, 5, 6, 7]], np.int32)
    res = np.array([[0, 2], [5, 4]], np.int32)
    e = orc.evaluate(None, None, gt, 2, res)
    assert e["c1"] == 0.5 and e["r_k_query"] == 1.0 and e["c_k_query"] == 0.75
    assert np.isnan(e["c1_dup"])


# ---- end-to-end oracle sanity ---------------------------------------------------------------
def test_build_and_query_recall(orc, small_graph):
    g = small_graph
    c = g["cfg"]
    graph0 = g["graph"][:g["N"]]
    assert graph0.min() >= 0 and graph0.max() < g["N"]
    # translation of layer l points into layer 0 ids; selection into layer l-1
    for l in (1, 2, 3):
        tr = g["tr"][c.STs_offsets[l]:c.STs_offsets[l] + c.Ns[l]]
        sel = g["sel"][c.STs_offsets[l]:c.STs_offsets[l] + c.Ns[l]]
        assert tr.min() >= 0 and tr.max() < g["N"] and len(set(tr)) == len(tr)
        assert sel.min() >= 0 and sel.max() < c.Ns[l - 1]
    q = make_int_data(100, g["D"], 4321)
    start = g["tr"][c.STs_offsets[3]:c.STs_offsets[3] + c.Ns[3]]
    ids, d, nd, npop = orc.query(g["base"], q, graph0, start, g["stats"], 10, 0.64, 400,
                                 counters=True)
    gt, _ = orc.bf_query(g["base"], q, 10)
    rec = np.mean([len(set(a) & set(b)) / 10 for a, b in zip(ids, gt)])
    assert rec > 0.97
    assert (np.diff(d, axis=1) >= 0).all()
    assert (npop <= 400).all() and (nd >= 32).all()


def test_oracle_build_regression_fixture(orc):
    """tests/golden/oracle_build.json: digests of orc.build on seeded data.  Guards the oracle --
    the anchor of every GPU parity test -- against accidental change; it pins nothing to the
    reference (which has no runnable build here)."""
    import importlib.util
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_oracle_build_golden",
                                                  os.path.join(here, "make_oracle_build_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for entry in json.load(open(os.path.join(here, "oracle_build.json"))):
        assert mod.digest(entry["case"]) == entry["digest"], entry["case"]


def test_wave_order_vs_reference_order(orc):
    """The oracle's two float summation orders (the reference's restated cub::BlockReduce order and
    the product kernels' DistEngine order, orc.wave_order): identical on integer-valued data, where
    every order is exact, and within the contract's 1e-4 on fractional data -- what the GPU
    parity tests, which pin the kernels bit for bit to the second order, leave open."""
    from conftest import make_int_data, make_uni_data
    from parity_helpers import assert_order_tolerance
    for D in (32, 96, 128, 960):
        for measure in (0, 1):
            bi, qi = make_int_data(400, D, 5), make_int_data(8, D, 6)
            ref = orc.bf_query(bi, qi, 10, measure)
            with orc.wave_order():
                wav = orc.bf_query(bi, qi, 10, measure)
            if measure == 0:
                assert np.array_equal(ref[0], wav[0]) and np.array_equal(ref[1], wav[1])
            bu, qu = make_uni_data(400, D, 7), make_uni_data(8, D, 8)
            ref = orc.bf_query(bu, qu, 10, measure)
            with orc.wave_order():
                wav = orc.bf_query(bu, qu, 10, measure)
            same = ref[0] == wav[0]
            assert same.mean() > 0.9
            assert_order_tolerance(wav[1], ref[1], same, D, measure, (D, measure))
    # uint8 rows: exact integers in both orders
    r = np.random.default_rng(3)
    b8 = r.integers(0, 256, (300, 128)).astype(np.uint8)
    q8 = r.integers(0, 256, (6, 128)).astype(np.uint8)
    ref = orc.bf_query(b8, q8, 10, 0)
    with orc.wave_order():
        wav = orc.bf_query(b8, q8, 10, 0)
    assert np.array_equal(ref[0], wav[0]) and np.array_equal(ref[1], wav[1])
