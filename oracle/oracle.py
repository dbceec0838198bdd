"""TEST INFRASTRUCTURE ONLY -- ctypes binding of the CPU oracle (oracle/ggnn_oracle.hpp).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (ggnn_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libggnn_oracle.so")

EUCLIDEAN, COSINE = 0, 1
F32, U8 = 0, 1


def build_lib(force=False):
    src = [os.path.join(_HERE, f) for f in ("ggnn_oracle.cpp", "ggnn_oracle.hpp", "wave_model.hpp")]
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libggnn_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


class GraphConfig(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in
                ("N", "D", "KBuild", "KF", "G", "S", "S0", "S0_off", "SG", "SG_off", "N_all",
                 "ST_all")] + [("Bs", C.c_uint32 * 4), ("Ns", C.c_uint32 * 4),
                               ("Ns_offsets", C.c_uint32 * 4), ("STs_offsets", C.c_uint32 * 4)]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if hasattr(v, "__len__") else int(v)
        return d


class QuerySizing(C.Structure):
    _fields_ = [("cache_size", C.c_uint32), ("sorted_size", C.c_uint32),
                ("block_dim_x", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build_lib()
        _lib = C.CDLL(_SO)
        _lib.orc_distance.restype = C.c_float
        _lib.orc_margin_min.restype = C.c_double
        _lib.orc_construction_block.restype = C.c_uint32
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _dtype_code(a):
    if a.dtype == np.float32:
        return F32
    if a.dtype == np.uint8:
        return U8
    raise TypeError("base/query must be float32 or uint8")


def _c(a, dtype=None):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


def graph_config(N, D, K):
    cfg = GraphConfig()
    lib().orc_graph_config(C.c_uint32(N), C.c_uint32(D), C.c_uint32(K), C.byref(cfg))
    return cfg


def query_sizing(D, KQuery, max_iters):
    s = QuerySizing()
    rc = lib().orc_query_sizing(C.c_uint32(D), C.c_uint32(KQuery), C.c_uint32(max_iters),
                                C.byref(s))
    if rc:
        raise ValueError("query parameters out of range")
    return s


def distance(base, query_row, other_id, measure=EUCLIDEAN, block=32, items=4):
    base = _c(base)
    query_row = _c(query_row)
    return float(lib().orc_distance(_p(base), _p(query_row), C.c_uint32(base.shape[1]),
                                    _dtype_code(base), measure, C.c_uint64(other_id),
                                    C.c_uint32(block), C.c_uint32(items)))


def bf_query(base, query, K, measure=EUCLIDEAN, threads=0):
    base, query = _c(base), _c(query)
    N, D = base.shape
    Nq = query.shape[0]
    ids = np.empty((Nq, K), np.int32)
    dists = np.empty((Nq, K), np.float32)
    lib().orc_bf_query(_p(base), C.c_uint32(N), C.c_uint32(D), _dtype_code(base), _p(query),
                       C.c_uint32(Nq), C.c_uint32(K), measure, _p(ids), _p(dists), threads)
    return ids, dists


def query(base, query, graph0, start, nn1_stats, KQuery, tau_query, max_iters=400,
          measure=EUCLIDEAN, shards_per_gpu=1, on_gpu_shard=0, out=None, threads=0,
          counters=False):
    base, query = _c(base), _c(query)
    graph0 = _c(graph0, np.int32)
    start = _c(start, np.int32)
    nn1_stats = _c(nn1_stats, np.float32)
    N, D = base.shape
    Nq = query.shape[0]
    KBuild = graph0.shape[1]
    if out is None:
        ids = np.full((Nq, KQuery * shards_per_gpu), -7, np.int32)
        dists = np.full((Nq, KQuery * shards_per_gpu), np.nan, np.float32)
    else:
        ids, dists = out
    nd = np.zeros(Nq, np.uint32)
    npop = np.zeros(Nq, np.uint32)
    lib().orc_query(_p(base), C.c_uint32(N), C.c_uint32(D), _dtype_code(base), _p(query),
                    C.c_uint32(Nq), _p(graph0), C.c_uint32(KBuild), _p(start),
                    C.c_uint32(start.size), _p(nn1_stats), C.c_uint32(KQuery),
                    C.c_float(tau_query), C.c_uint32(max_iters), measure,
                    C.c_uint32(shards_per_gpu), C.c_uint32(on_gpu_shard), _p(ids), _p(dists),
                    _p(nd), _p(npop), threads)
    if counters:
        return ids, dists, nd, npop
    return ids, dists


def top(base, KBuild, translation, Nlayer, S, S_offset, layer, measure=EUCLIDEAN, threads=0):
    base = _c(base)
    graph = np.empty((Nlayer, KBuild), np.int32)
    nn1 = np.empty(Nlayer, np.float32)
    tr = None if translation is None else _c(translation, np.int32)
    lib().orc_top(_p(base), C.c_uint32(base.shape[1]), _dtype_code(base), measure,
                  C.c_uint32(KBuild), _p(tr), C.c_uint32(Nlayer), C.c_uint32(S),
                  C.c_uint32(S_offset), C.c_uint32(layer), _p(graph), _p(nn1), threads)
    return graph, nn1


def select(cfg, layer, nn1_dist_buffer, rng, translation_all, selection_all):
    """in-place on translation_all / selection_all ([ST_all] int32)"""
    nn1 = _c(nn1_dist_buffer, np.float32)
    rng = _c(rng, np.float32)
    assert translation_all.dtype == np.int32 and selection_all.dtype == np.int32
    lib().orc_select(C.byref(cfg), C.c_uint32(layer), _p(nn1), _p(rng), _p(translation_all),
                     _p(selection_all))


def merge(base, cfg, graph_all, translation_all, selection_all, nn1_stats, tau_build, layer_top,
          layer_btm, measure=EUCLIDEAN, threads=0, counters=False):
    base = _c(base)
    graph_all = _c(graph_all, np.int32)
    translation_all = _c(translation_all, np.int32)
    selection_all = _c(selection_all, np.int32)
    nn1_stats = _c(nn1_stats, np.float32)
    Nb = cfg.Ns[layer_btm]
    gb = np.empty((Nb, cfg.KBuild), np.int32)
    nn1 = np.zeros(Nb, np.float32)
    nd = np.zeros(Nb, np.uint32)
    lib().orc_merge(_p(base), _dtype_code(base), measure, C.byref(cfg), _p(graph_all),
                    _p(translation_all), _p(selection_all), _p(nn1_stats), C.c_float(tau_build),
                    C.c_uint32(layer_top), C.c_uint32(layer_btm), _p(gb), _p(nn1), _p(nd),
                    threads)
    if counters:
        return gb, nn1, nd
    return gb, nn1


def sym(base, KBuild, graph_layer, translation_layer, nn1_stats, tau_build, sym_buffer,
        sym_atomic, measure=EUCLIDEAN, first_n=0, count=None):
    """in-place on sym_buffer [N x KF] int32 / sym_atomic [N] uint32"""
    base = _c(base)
    graph_layer = _c(graph_layer, np.int32)
    tr = None if translation_layer is None else _c(translation_layer, np.int32)
    nn1_stats = _c(nn1_stats, np.float32)
    Nl = graph_layer.shape[0]
    if count is None:
        count = Nl
    assert sym_buffer.dtype == np.int32 and sym_atomic.dtype == np.uint32
    lib().orc_sym(_p(base), _dtype_code(base), measure, C.c_uint32(base.shape[1]),
                  C.c_uint32(KBuild), _p(graph_layer), _p(tr), C.c_uint32(Nl), _p(nn1_stats),
                  C.c_float(tau_build), _p(sym_buffer), _p(sym_atomic), C.c_uint32(first_n),
                  C.c_uint32(count))


def sym_buffer_merge(KBuild, sym_buffer, sym_atomic, graph_layer):
    """in-place on graph_layer [N x K] int32"""
    assert graph_layer.dtype == np.int32 and graph_layer.flags.c_contiguous
    lib().orc_sym_buffer_merge(C.c_uint32(KBuild), C.c_uint32(graph_layer.shape[0]),
                               _p(_c(sym_buffer, np.int32)), _p(_c(sym_atomic, np.uint32)),
                               _p(graph_layer))


def nn1_stats(nn1_dist_buffer):
    v = _c(nn1_dist_buffer, np.float32)
    out = np.zeros(2, np.float32)
    lib().orc_nn1_stats(_p(v), C.c_uint32(v.size), _p(out))
    return out


def make_rng(N, seed=1234):
    """[4 x N] uniform (0,1] float32 as select() expects (stand-in for cuRAND, unpinned)."""
    g = np.random.Generator(np.random.PCG64(seed))
    return (1.0 - g.random((4, N), dtype=np.float32)).astype(np.float32)


def build(base, KBuild, tau_build, refinement_iterations=2, measure=EUCLIDEAN, rng=None,
          threads=0):
    base = _c(base)
    N, D = base.shape
    cfg = graph_config(N, D, KBuild)
    if rng is None:
        rng = make_rng(N)
    rng = _c(rng, np.float32)
    graph_all = np.full((cfg.N_all, KBuild), -1, np.int32)
    tr = np.full(cfg.ST_all, -1, np.int32)
    sel = np.full(cfg.ST_all, -1, np.int32)
    stats = np.zeros(2, np.float32)
    lib().orc_build(_p(base), _dtype_code(base), measure, C.byref(cfg), C.c_float(tau_build),
                    C.c_uint32(refinement_iterations), _p(rng), _p(graph_all), _p(tr), _p(sel),
                    _p(stats), threads)
    return cfg, graph_all, tr, sel, stats


def sort_shard_results(ids, dists):
    ids = np.array(ids, np.int32, order="C")
    dists = np.array(dists, np.float32, order="C")
    lib().orc_sort_shard_results(C.c_uint32(ids.shape[0]), C.c_uint32(ids.shape[1]), _p(ids),
                                 _p(dists))
    return ids, dists


def merge_results(part_ids, part_dists, K, shards_per_gpu, N_shard):
    G = len(part_ids)
    part_ids = [_c(a, np.int32) for a in part_ids]
    part_dists = [_c(a, np.float32) for a in part_dists]
    Nq = part_ids[0].shape[0]
    pi = (C.c_void_p * G)(*[a.ctypes.data for a in part_ids])
    pd = (C.c_void_p * G)(*[a.ctypes.data for a in part_dists])
    ids = np.empty((Nq, K), np.int32)
    dists = np.empty((Nq, K), np.float32)
    lib().orc_merge_results(C.c_uint32(Nq), C.c_uint32(K), C.c_uint32(G),
                            C.c_uint32(shards_per_gpu), C.c_uint32(N_shard), pi, pd, _p(ids),
                            _p(dists))
    return ids, dists


def evaluate(base, query, gt, KQuery, results, measure=EUCLIDEAN):
    gt = _c(gt, np.int32)
    results = _c(results, np.int32)
    out = np.zeros(7, np.float32)
    if base is not None:
        base, query = _c(base), _c(query)
        args = (_p(base), C.c_uint32(base.shape[0]), _p(query), C.c_uint32(query.shape[0]),
                C.c_uint32(base.shape[1]), _dtype_code(base))
    else:
        args = (None, C.c_uint32(0), None, C.c_uint32(0), C.c_uint32(0), 0)
    lib().orc_evaluate(*args, measure, _p(gt), C.c_uint32(gt.shape[1]), C.c_uint32(KQuery),
                       _p(results), C.c_uint32(results.shape[0]), _p(out))
    names = ("c1", "c1_dup", "c_k_query", "c_k_query_dup", "r_k_query", "r_k_query_dup")
    return dict(zip(names, (float(x) for x in out[:6])))


def _script(fn, BEST, SORTED, CACHE, xi, ops, BLOCK=None):
    ops = np.ascontiguousarray(ops, np.int32).reshape(-1, 3)
    keys = np.empty(CACHE, np.int32)
    dists = np.empty(SORTED, np.float32)
    pops = np.empty(len(ops), np.int32)
    heads = np.zeros(2, np.uint32)
    if BLOCK is None:
        fn(C.c_uint32(BEST), C.c_uint32(SORTED), C.c_uint32(CACHE), C.c_float(xi), _p(ops),
           C.c_uint32(len(ops)), _p(keys), _p(dists), _p(pops), _p(heads))
    else:
        fn(C.c_uint32(BEST), C.c_uint32(SORTED), C.c_uint32(CACHE), C.c_uint32(BLOCK),
           C.c_float(xi), _p(ops), C.c_uint32(len(ops)), _p(keys), _p(dists), _p(pops),
           _p(heads))
    return keys, dists, pops, heads


def kbest_script(BEST, BLOCK, dists, ids, check_worst=True):
    dists = _c(dists, np.float32)
    ids = _c(ids, np.int32)
    out_d = np.empty(BEST, np.float32)
    out_i = np.empty(BEST, np.int32)
    lib().orc_kbest_script(C.c_uint32(BEST), C.c_uint32(BLOCK), _p(dists), _p(ids),
                           C.c_uint32(dists.size), int(bool(check_worst)), _p(out_d), _p(out_i))
    return out_d, out_i


REF_KBEST_SO = os.path.join(_HERE, "_ref", "libggnn_ref_kbest.so")
REF_DEF_SO = os.path.join(_HERE, "_ref", "libggnn_ref_def.so")


def bit_ceil(v):
    f = lib().orc_bit_ceil
    f.restype = C.c_uint32
    return int(f(C.c_uint32(v)))


def next_multiple32(v):
    f = lib().orc_next_multiple32
    f.restype = C.c_uint32
    return int(f(C.c_uint32(v)))


def align8(v):
    f = lib().orc_align8
    f.restype = C.c_size_t
    return int(f(C.c_size_t(v)))


def cache_script(BEST, SORTED, CACHE, BLOCK, xi, ops):
    return _script(lib().orc_cache_script, BEST, SORTED, CACHE, xi, ops, BLOCK)


def wave_model_script(BEST, SORTED, CACHE, xi, ops):
    return _script(lib().orc_wave_model_script, BEST, SORTED, CACHE, xi, ops)


def op_push(key, dist):
    return (0, int(key), int(np.float32(dist).view(np.int32)))


def op_pop():
    return (1, 0, 0)


def op_xi(x):
    return (2, 0, int(np.float32(x).view(np.int32)))


def op_transform():
    return (3, 0, 0)


def set_wave_order(enable):
    """float sums in the product kernels' order (bit-for-bit comparisons on float data) instead
    of the reference's restated cub::BlockReduce order"""
    lib().orc_set_wave_order(int(bool(enable)))


class wave_order:
    def __enter__(self):
        set_wave_order(True)

    def __exit__(self, *exc):
        set_wave_order(False)
        return False


def set_fast_distance(enable):
    lib().orc_set_fast_distance(int(bool(enable)))


def margin_reset():
    lib().orc_margin_reset()


def margin_min():
    return float(lib().orc_margin_min())
