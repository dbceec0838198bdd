"""Operator seam of the engine on torch CUDA tensors (include/ggnn_c.h section 2).

Mirrors the reference's internal operator interface -- QueryKernels::{query, bruteForceQuery}
(include/ggnn/query/query_kernels.cuh:47-57) and the kernels GraphConstruction drives
(src/ggnn/construction/graph_construction.cu:128-379) -- one function per kernel, all on the
current torch stream.  torch is only the owner of the device memory here.
"""
import torch

from . import _lib
from ._lib import GraphConfig, check, lib

EUCLIDEAN, COSINE = 0, 1


def _dtype_code(t):
    if t.dtype == torch.float32:
        return _lib.F32
    if t.dtype == torch.uint8:
        return _lib.U8
    raise TypeError("base/query must be float32 or uint8")


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need(t, dtype=None, name="tensor"):
    if not t.is_cuda or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous CUDA tensor")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}")
    return t


def _cfg(cfg):
    """accept any structure with the ggnn_graph_config fields"""
    if isinstance(cfg, GraphConfig):
        return cfg
    out = GraphConfig()
    for name, _ in GraphConfig._fields_:
        v = getattr(cfg, name)
        if hasattr(v, "__len__"):
            for i in range(4):
                getattr(out, name)[i] = v[i]
        else:
            setattr(out, name, v)
    return out


def graph_config(N, D, KBuild):
    cfg = GraphConfig()
    check(lib().ggnn_graph_config_init(N, D, KBuild, cfg))
    return cfg


def query_sizing(D, k_query, max_iterations):
    import ctypes as C
    cache, sorted_ = C.c_uint32(), C.c_uint32()
    check(lib().ggnn_query_sizing(D, k_query, max_iterations, C.byref(cache), C.byref(sorted_)))
    return cache.value, sorted_.value


def prescreen_sizes(N, D, measure=EUCLIDEAN):
    import ctypes as C
    dc, pf, sf = C.c_uint32(), C.c_size_t(), C.c_size_t()
    check(lib().ggnn_prescreen_sizes(N, D, measure, C.byref(dc), C.byref(pf), C.byref(sf)))
    return dc.value, pf.value, sf.value


def prescreen_encode(base, measure=EUCLIDEAN):
    """8-bit pre-screen copy of a float32 base for `measure`:
    (codes [N, code_dim] uint8, params float32)."""
    _need(base, torch.float32, "base")
    dc, pf, sf = prescreen_sizes(base.shape[0], base.shape[1], measure)
    codes = torch.empty((base.shape[0], dc), dtype=torch.uint8, device=base.device)
    params = torch.empty(pf, dtype=torch.float32, device=base.device)
    scratch = torch.empty(sf, dtype=torch.float32, device=base.device)
    check(lib().ggnn_op_prescreen_encode(_ptr(base), base.shape[0], base.shape[1], measure,
                                         _ptr(codes), _ptr(params), _ptr(scratch), _stream()))
    return codes, params


def prescreen_probe(codes, params, query, cand, crit, measure=EUCLIDEAN):
    """reject [Nq, M] int32 and coded squared distances [Nq, M] for explicit triples."""
    _need(codes, torch.uint8, "codes"), _need(params, torch.float32, "params")
    _need(query, torch.float32, "query"), _need(cand, torch.int32, "cand")
    _need(crit, torch.float32, "crit")
    Nq, M = cand.shape
    reject = torch.empty((Nq, M), dtype=torch.int32, device=codes.device)
    s_out = torch.empty((Nq, M), dtype=torch.float32, device=codes.device)
    check(lib().ggnn_op_prescreen_probe(_ptr(codes), _ptr(params), query.shape[1], measure,
                                        _ptr(query), Nq, _ptr(cand), M, _ptr(crit), _ptr(reject),
                                        _ptr(s_out), _stream()))
    return reject, s_out


def query(base, query, graph0, start, nn1_stats, k_query, tau_query, max_iterations=400,
          measure=EUCLIDEAN, shards_per_gpu=1, on_gpu_shard=0, out=None, counters=False,
          prescreen=None, rows_read=None):
    """prescreen: optional (codes, params) of prescreen_encode(base, measure) (float32);
    rows_read: optional int32 [Nq, 2] tensor receiving the float / code rows read per query."""
    _need(base, name="base"), _need(query, base.dtype, "query")
    _need(graph0, torch.int32, "graph0"), _need(start, torch.int32, "start")
    _need(nn1_stats, torch.float32, "nn1_stats")
    Nq = query.shape[0]
    if out is None:
        ids = torch.empty((Nq, k_query * shards_per_gpu), dtype=torch.int32, device=base.device)
        dists = torch.empty((Nq, k_query * shards_per_gpu), dtype=torch.float32,
                            device=base.device)
    else:
        ids, dists = out
    nd = npop = None
    if counters:
        nd = torch.zeros(Nq, dtype=torch.int32, device=base.device)
        npop = torch.zeros(Nq, dtype=torch.int32, device=base.device)
    if prescreen is not None:
        codes, params = prescreen
        _need(base, torch.float32, "base"), _need(codes, torch.uint8, "codes")
        _need(params, torch.float32, "params")
        check(lib().ggnn_op_query_prescreened(
            _ptr(base), base.shape[0], base.shape[1], _ptr(codes), _ptr(params), _ptr(query), Nq,
            _ptr(graph0), graph0.shape[1], _ptr(start), start.numel(), _ptr(nn1_stats), k_query,
            tau_query, max_iterations, measure, shards_per_gpu, on_gpu_shard, _ptr(ids),
            _ptr(dists),
            _ptr(nd), _ptr(npop), _ptr(rows_read), _stream()))
    else:
        check(lib().ggnn_op_query(_ptr(base), _dtype_code(base), base.shape[0], base.shape[1],
                                  _ptr(query), Nq, _ptr(graph0), graph0.shape[1], _ptr(start),
                                  start.numel(), _ptr(nn1_stats), k_query, tau_query,
                                  max_iterations, measure, shards_per_gpu, on_gpu_shard,
                                  _ptr(ids), _ptr(dists), _ptr(nd), _ptr(npop), _stream()))
    if counters:
        return ids, dists, nd, npop
    return ids, dists


def bf_query(base, query, k_query, measure=EUCLIDEAN, rescanned=False):
    """rescanned=True: also return how many queries the matrix-core path handed to the scan"""
    _need(base, name="base"), _need(query, base.dtype, "query")
    Nq = query.shape[0]
    ids = torch.empty((Nq, k_query), dtype=torch.int32, device=base.device)
    dists = torch.empty((Nq, k_query), dtype=torch.float32, device=base.device)
    if rescanned:
        n = torch.zeros(1, dtype=torch.int32, device=base.device)
        check(lib().ggnn_op_bf_query_certified(_ptr(base), _dtype_code(base), base.shape[0],
                                               base.shape[1], _ptr(query), Nq, k_query, measure,
                                               _ptr(ids), _ptr(dists), _ptr(n), _stream()))
        return ids, dists, int(n.item())
    check(lib().ggnn_op_bf_query(_ptr(base), _dtype_code(base), base.shape[0], base.shape[1],
                                 _ptr(query), Nq, k_query, measure, _ptr(ids), _ptr(dists),
                                 _stream()))
    return ids, dists


def top(base, KBuild, translation_layer, N_layer, S, S_offset, layer, measure=EUCLIDEAN):
    _need(base, name="base")
    graph = torch.empty((N_layer, KBuild), dtype=torch.int32, device=base.device)
    nn1 = torch.empty(N_layer, dtype=torch.float32, device=base.device)
    check(lib().ggnn_op_top(_ptr(base), _dtype_code(base), base.shape[1], measure, KBuild,
                            _ptr(translation_layer), N_layer, S, S_offset, layer, _ptr(graph),
                            _ptr(nn1), _stream()))
    return graph, nn1


def merge(base, cfg, graph_all, translation_all, selection_all, nn1_stats, tau_build, layer_top,
          layer_btm, measure=EUCLIDEAN, counters=False, prescreen=None):
    _need(base, name="base"), _need(graph_all, torch.int32, "graph_all")
    _need(translation_all, torch.int32), _need(selection_all, torch.int32)
    Nb = cfg.Ns[layer_btm]
    gb = torch.empty((Nb, cfg.KBuild), dtype=torch.int32, device=base.device)
    nn1 = torch.zeros(Nb, dtype=torch.float32, device=base.device)
    nd = torch.zeros(Nb, dtype=torch.int32, device=base.device) if counters else None
    if prescreen is not None:
        codes, params = prescreen
        _need(base, torch.float32, "base"), _need(codes, torch.uint8, "codes")
        _need(params, torch.float32, "params")
        check(lib().ggnn_op_merge_prescreened(
            _ptr(base), _ptr(codes), _ptr(params), measure, _cfg(cfg), _ptr(graph_all),
            _ptr(translation_all), _ptr(selection_all), _ptr(nn1_stats), tau_build, layer_top,
            layer_btm, _ptr(gb), _ptr(nn1), _ptr(nd), _stream()))
    else:
        check(lib().ggnn_op_merge(_ptr(base), _dtype_code(base), measure, _cfg(cfg),
                                  _ptr(graph_all), _ptr(translation_all), _ptr(selection_all),
                                  _ptr(nn1_stats), tau_build, layer_top, layer_btm, _ptr(gb),
                                  _ptr(nn1), _ptr(nd), _stream()))
    if counters:
        return gb, nn1, nd
    return gb, nn1


def select(cfg, layer, nn1_dist_buffer, rng, translation_all, selection_all):
    check(lib().ggnn_op_select(_cfg(cfg), layer, _ptr(nn1_dist_buffer), _ptr(rng),
                               _ptr(translation_all), _ptr(selection_all), _stream()))


def uniform(n, seed=1234, stream_id=0, device="cuda"):
    out = torch.empty(n, dtype=torch.float32, device=device)
    check(lib().ggnn_op_uniform(_ptr(out), n, seed, stream_id, _stream()))
    return out


def sym(base, KBuild, graph_layer, translation_layer, nn1_stats, tau_build, sym_buffer,
        sym_atomic, measure=EUCLIDEAN, first_n=0, count=None, prescreen=None):
    """prescreen: optional (codes, params) of prescreen_encode(base, measure) (float32)"""
    N_layer = graph_layer.shape[0]
    if count is None:
        count = N_layer
    if prescreen is not None:
        codes, params = prescreen
        check(lib().ggnn_op_sym_prescreened(_ptr(base), _ptr(codes), _ptr(params), measure,
                                            base.shape[1], KBuild, _ptr(graph_layer),
                                            _ptr(translation_layer), N_layer, _ptr(nn1_stats),
                                            tau_build, _ptr(sym_buffer), _ptr(sym_atomic), first_n,
                                            count, _stream()))
        return
    check(lib().ggnn_op_sym(_ptr(base), _dtype_code(base), measure, base.shape[1], KBuild,
                            _ptr(graph_layer), _ptr(translation_layer), N_layer, _ptr(nn1_stats),
                            tau_build, _ptr(sym_buffer), _ptr(sym_atomic), first_n, count,
                            _stream()))


def sym_buffer_merge(KBuild, sym_buffer, sym_atomic, graph_layer):
    check(lib().ggnn_op_sym_buffer_merge(KBuild, graph_layer.shape[0], _ptr(sym_buffer),
                                         _ptr(sym_atomic), _ptr(graph_layer), _stream()))


def nn1_stats(nn1_dist_buffer):
    scratch = torch.empty(lib().ggnn_nn1_stats_scratch_floats(), dtype=torch.float32,
                          device=nn1_dist_buffer.device)
    out = torch.empty(2, dtype=torch.float32, device=nn1_dist_buffer.device)
    check(lib().ggnn_op_nn1_stats(_ptr(nn1_dist_buffer), nn1_dist_buffer.numel(), _ptr(scratch),
                                  _ptr(out), _stream()))
    return out


def sort_shard_results(ids, dists):
    """in place"""
    check(lib().ggnn_op_sort_shard_results(ids.shape[0], ids.shape[1], _ptr(ids), _ptr(dists),
                                           _stream()))
    return ids, dists


def merge_results(parts_ids, parts_dists, k, id_offset_per_part):
    """parts_*: [num_parts, Nq, stride] (e.g. the all-gather buffer)"""
    _need(parts_ids, torch.int32), _need(parts_dists, torch.float32)
    P, Nq, stride = parts_ids.shape
    ids = torch.empty((Nq, k), dtype=torch.int32, device=parts_ids.device)
    dists = torch.empty((Nq, k), dtype=torch.float32, device=parts_ids.device)
    check(lib().ggnn_op_merge_results(Nq, k, P, stride, id_offset_per_part, _ptr(parts_ids),
                                      _ptr(parts_dists), _ptr(ids), _ptr(dists), _stream()))
    return ids, dists
