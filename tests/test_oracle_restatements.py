"""Second, independent restatements (numpy, written from the cited reference lines, sharing no
code with oracle/ggnn_oracle.cpp) of the build kernels that have no pinned reference build:
`top` (top_merge_layer.cu:40-82, quirk Q4), the WRS selection (wrs_select_layer.cu:41-102 with
the launch parameters of graph_construction.cu:163-187), `sym_buffer_merge`
(sym_buffer_merge_layer.cu:36-99) and the nn1 statistics (graph_construction.cu:381-402).  Two
derivations that agree bit for bit on seeded inputs are the strongest pin this image allows for
these rows (the reference has no tests and its kernels need nvcc + CUB).

Integer data keeps every squared L2 distance exact in float32, so no summation order is involved.
"""
import ctypes
import ctypes.util

import numpy as np
import pytest

_libm = ctypes.CDLL(ctypes.util.find_library("m"))
_libm.logf.restype = ctypes.c_float
_libm.logf.argtypes = [ctypes.c_float]
_libm.sqrtf.restype = ctypes.c_float
_libm.sqrtf.argtypes = [ctypes.c_float]


def int_data(N, D, seed):
    return np.random.default_rng(seed).integers(0, 256, (N, D)).astype(np.float32)


def l2sq(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    return np.float32((d * d).sum())  # exact: integers, < 2^24


# ---- top_merge_layer.cu:40-82 ------------------------------------------------------------------
def np_top(base, KBuild, translation, Nlayer, S, S_offset, layer):
    graph = np.empty((Nlayer, KBuild), np.int32)
    nn1 = np.empty(Nlayer, np.float32)
    s_plus = S_offset * (S + 1)                                     # :52
    for n in range(Nlayer):
        m = n if layer == 0 else int(translation[n])                # :47
        s_act = S + 1 if (layer == 0 and n < s_plus) else S         # :53
        if layer or n < s_plus:                                     # :55-57
            start = (n // s_act) * s_act
        else:
            start = s_plus + ((n - s_plus) // s_act) * s_act
        cand = []
        for other_n in range(start, start + s_act):                 # :60-69
            other_m = int(translation[other_n]) if layer else other_n
            if other_m == m:
                continue
            cand.append((l2sq(base[m], base[other_m]), other_n))
        # add_unique keeps equal distances in insertion order (k_best_list.cuh:92-103): stable
        order = sorted(range(len(cand)), key=lambda i: cand[i][0])
        ids = [cand[i][1] for i in order][:KBuild]
        graph[n] = ids
        nn1[n] = _libm.sqrtf(cand[order[1]][0])                     # :76-80: s_dists[1], sqrt (Q4)
    return graph, nn1


@pytest.mark.parametrize("N,K", [(1000, 24), (777, 20)])
def test_top_layer0_equals_numpy_restatement(orc, N, K):
    base = int_data(N, 32, 5)
    cfg = orc.graph_config(N, 32, K)
    g, nn1 = orc.top(base, K, None, N, cfg.S0, cfg.S0_off, 0)
    g2, nn2 = np_top(base, K, None, N, cfg.S0, cfg.S0_off, 0)
    assert np.array_equal(g, g2)
    assert nn1.tobytes() == nn2.tobytes()


def test_top_upper_layer_with_translation_equals_numpy_restatement(orc):
    """layer > 0: ids are layer-local, points come through the translation (a random injective map
    here), segments have exactly S points"""
    N, K = 4000, 24
    base = int_data(N, 16, 6)
    cfg = orc.graph_config(N, 16, K)
    Nl = (N // 7 // cfg.S) * cfg.S
    tr = np.random.default_rng(7).permutation(N)[:Nl].astype(np.int32)
    g, nn1 = orc.top(base, K, tr, Nl, cfg.S, 0, 1)
    g2, nn2 = np_top(base, K, tr, Nl, cfg.S, 0, 1)
    assert np.array_equal(g, g2)
    assert nn1.tobytes() == nn2.tobytes()


def test_top_with_ties_keeps_insertion_order(orc):
    """few distinct values: many equal distances inside a segment (quirk Q2)"""
    N, K = 500, 24
    base = np.random.default_rng(8).integers(0, 2, (N, 8)).astype(np.float32)
    cfg = orc.graph_config(N, 8, K)
    g, nn1 = orc.top(base, K, None, N, cfg.S0, cfg.S0_off, 0)
    g2, nn2 = np_top(base, K, None, N, cfg.S0, cfg.S0_off, 0)
    assert np.array_equal(g, g2)
    assert nn1.tobytes() == nn2.tobytes()


# ---- wrs_select_layer.cu:41-102 + graph_construction.cu:163-187 -----------------------------------
def np_select(cfg, layer, nn1, rng, translation_all, selection_all):
    S = cfg.S if layer else cfg.S0
    S_offset = 0 if layer else cfg.S0_off
    eps = np.float32(np.finfo(np.float32).eps)
    tr_layer = translation_all[cfg.STs_offsets[layer]:]
    dst = cfg.STs_offsets[layer + 1]
    for b in range(cfg.Bs[layer]):
        s_cur = S + (1 if b < S_offset else 0)                       # :50
        start = b * S + min(b, S_offset)                             # :51
        keys = []
        for i in range(s_cur):                                       # :56-67
            n = start + i
            num = np.float32(-1) * np.float32(_libm.logf(rng[n]))
            keys.append((np.float32(num / np.float32(nn1[n] + eps)), n))
        # stable descending radix sort (:76): equal keys keep their order
        order = sorted(range(s_cur), key=lambda i: -float(keys[i][0]))
        upper = b // cfg.G                                           # :79
        nth = b - upper * cfg.G                                      # :81
        count = cfg.SG + (1 if nth < cfg.SG_off else 0)              # :86
        dest = upper * cfg.S + nth * cfg.SG + min(nth, cfg.SG_off)   # :90-91
        for s in range(count):                                       # :95-102
            n = keys[order[s]][1]
            selection_all[dst + dest + s] = n
            translation_all[dst + dest + s] = n if layer == 0 else tr_layer[n]


@pytest.mark.parametrize("N", [6000, 20000, 33333])
def test_select_all_layers_equal_numpy_restatement(orc, N):
    """three selections in a row (layer 0 -> 1 -> 2 -> 3), each reading the previous layer's
    translation; shapes with S0_off > 0 and with SG_off > 0"""
    K = 24
    cfg = orc.graph_config(N, 16, K)
    rs = np.random.default_rng(N)
    tr_a = np.full(cfg.ST_all, -1, np.int32)
    sel_a = np.full(cfg.ST_all, -1, np.int32)
    tr_b, sel_b = tr_a.copy(), sel_a.copy()
    for layer in range(3):
        nl = cfg.Ns[layer]
        nn1 = (rs.random(nl, dtype=np.float32) * 100 + 1).astype(np.float32)
        rng = (1.0 - rs.random(nl, dtype=np.float32)).astype(np.float32)  # (0, 1]
        orc.select(cfg, layer, nn1, rng, tr_a, sel_a)
        np_select(cfg, layer, nn1, rng, tr_b, sel_b)
        assert np.array_equal(sel_a, sel_b), layer
        assert np.array_equal(tr_a, tr_b), layer
    top = slice(cfg.STs_offsets[3], cfg.STs_offsets[3] + cfg.Ns[3])
    assert (tr_a[top] >= 0).all() and len(set(tr_a[top].tolist())) == cfg.Ns[3]


def test_select_equal_keys_keep_segment_order(orc):
    """identical rng and nn1 everywhere: the stable sort must select the first points of each
    segment"""
    N, K = 6000, 24
    cfg = orc.graph_config(N, 16, K)
    tr_a = np.full(cfg.ST_all, -1, np.int32)
    sel_a = np.full(cfg.ST_all, -1, np.int32)
    tr_b, sel_b = tr_a.copy(), sel_a.copy()
    nn1 = np.full(N, 3.0, np.float32)
    rng = np.full(N, 0.5, np.float32)
    orc.select(cfg, 0, nn1, rng, tr_a, sel_a)
    np_select(cfg, 0, nn1, rng, tr_b, sel_b)
    assert np.array_equal(sel_a, sel_b) and np.array_equal(tr_a, tr_b)


# ---- sym_buffer_merge_layer.cu:36-99 ------------------------------------------------------------
def np_sym_buffer_merge(KBuild, sym_buffer, sym_atomic, graph):
    KF = KBuild // 2
    KL = KBuild - KF
    for n in range(graph.shape[0]):
        buf = sym_buffer[n].copy()
        num = int(sym_atomic[n])                                      # :56
        gb = graph[n, KL:KL + KF].copy()                              # :62
        for i in range(KF):                                           # :65-91
            found = num >= KF                                         # :68
            if not found:
                found = bool((buf == gb[i]).any())                    # :75-82 (all KF slots)
            if not found:
                buf[num] = gb[i]                                      # :87-90
                num += 1
        graph[n, KL:KL + KF] = np.where(buf >= 0, buf, n)             # :97-98


@pytest.mark.parametrize("K", [24, 20])
def test_sym_buffer_merge_equals_numpy_restatement(orc, K):
    N = 3000
    KF = K // 2
    rs = np.random.default_rng(K)
    graph = rs.integers(0, N, (N, K)).astype(np.int32)
    # requested inverse links: 0..KF+3 requests per point (the counter may run past the capacity)
    atom = rs.integers(0, KF + 4, N).astype(np.uint32)
    buf = np.full((N, KF), -1, np.int32)
    for n in range(N):
        c = min(int(atom[n]), KF)
        # some requests repeat an existing foreign link, some repeat each other
        pool = np.concatenate([graph[n, K - KF:], rs.integers(0, N, KF)])
        buf[n, :c] = rs.choice(pool, c)
    g1 = graph.copy()
    g2 = graph.copy()
    orc.sym_buffer_merge(K, buf, atom, g1)
    np_sym_buffer_merge(K, buf, atom, g2)
    assert np.array_equal(g1, g2)
    assert np.array_equal(g1[:, :K - KF], graph[:, :K - KF])  # own links untouched


# ---- computeNN1Stats, graph_construction.cu:381-402 -----------------------------------------------
def test_nn1_stats_mean_and_max(orc):
    v = (np.random.default_rng(3).random(100_000, dtype=np.float32) * 50).astype(np.float32)
    out = orc.nn1_stats(v)
    # float64 accumulation (documented deviation: the reference's cub::DeviceReduce order is
    # unspecified); the maximum is exact
    assert out[0] == np.float32(v.astype(np.float64).sum() / v.size)
    assert out[1] == v.max()


# ---- the build / refine schedule, graph_construction.cu:128-147, 186-201, 298-379, 381-402 --------
def py_schedule(orc, base, K, tau, refine, rng):
    """the launch order written from the cited lines, driving the oracle's per-kernel entry points
    from Python; `orc.build` (one C++ function, the thing the engine's whole build is compared
    with) must produce the same arrays"""
    N, D = base.shape
    cfg = orc.graph_config(N, D, K)
    KF = K // 2
    L = 4
    graph_all = np.full((cfg.N_all, K), -1, np.int32)
    tr = np.full(cfg.ST_all, -1, np.int32)
    sel = np.full(cfg.ST_all, -1, np.int32)
    nn1_buf = np.zeros(N, np.float32)
    stats = np.zeros(2, np.float32)

    def layer_rows(l):
        return graph_all[cfg.Ns_offsets[l]:cfg.Ns_offsets[l] + cfg.Ns[l]]

    def layer_tr(l):
        return None if l == 0 else tr[cfg.STs_offsets[l]:cfg.STs_offsets[l] + cfg.Ns[l]]

    def merge(top, btm):                                             # :186-201
        nonlocal stats
        if top == btm:                                               # top(), :203-240
            g, nn1 = orc.top(base, K, layer_tr(btm), cfg.Ns[btm], cfg.S if btm else cfg.S0,
                             0 if btm else cfg.S0_off, btm)
            layer_rows(btm)[:] = g
            nn1_buf[:cfg.Ns[btm]] = nn1
        else:                                                        # mergeLayer, :242-296
            gb, nn1 = orc.merge(base, cfg, graph_all, tr, sel, stats, tau, top, btm)
            layer_rows(btm)[:] = gb                                  # graph_buffer -> graph, :292-295
            if btm == 0:
                nn1_buf[:] = nn1
        if btm == 0:                                                 # computeNN1Stats, :381-402
            stats = orc.nn1_stats(nn1_buf)

    def sym(layer):                                                  # :298-379
        buf = np.full((cfg.Ns[layer], KF), -1, np.int32)
        atom = np.zeros(cfg.Ns[layer], np.uint32)
        rows = np.ascontiguousarray(layer_rows(layer))
        orc.sym(base, K, rows, layer_tr(layer), stats, tau, buf, atom)
        orc.sym_buffer_merge(K, buf, atom, rows)
        layer_rows(layer)[:] = rows

    for top in range(L):                                             # build, :128-140
        for btm in range(top, -1, -1):
            merge(top, btm)
            if top < L - 1 and top == btm:
                orc.select(cfg, top, nn1_buf, rng[top], tr, sel)      # :163-187
            sym(btm)
    for _ in range(refine):                                          # refine, :141-147
        for layer in range(L - 2, -1, -1):
            merge(L - 1, layer)
            sym(layer)
    return cfg, graph_all, tr, sel, stats


@pytest.mark.parametrize("N,D,K,refine", [(3000, 32, 24, 2), (6000, 16, 20, 1), (2048, 64, 24, 0)])
def test_build_schedule_equals_python_schedule_over_oracle_kernels(orc, N, D, K, refine):
    base = (np.random.default_rng(N).integers(0, 52, (N, D)) * 5).astype(np.float32)
    rng = orc.make_rng(N, 17)
    cfg, graph, tr, sel, stats = orc.build(base, K, 0.5, refine, rng=rng)
    cfg2, graph2, tr2, sel2, stats2 = py_schedule(orc, base, K, 0.5, refine, rng)
    assert np.array_equal(tr, tr2), "translation"
    assert np.array_equal(sel, sel2), "selection"
    assert stats.tobytes() == np.asarray(stats2, np.float32).tobytes(), (stats, stats2)
    for l in range(4):
        a, b = cfg.Ns_offsets[l], cfg.Ns_offsets[l] + cfg.Ns[l]
        assert np.array_equal(graph[a:b], graph2[a:b]), f"layer {l}"
