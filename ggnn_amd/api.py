"""Python surface of the engine: same names, argument order and defaults as the reference's
nanobind module (src/ggnn/python/nanobind.cu:131-301), implemented over the C-ABI.

    import ggnn_amd as ggnn            # or `import ggnn` through the shim package
    g = ggnn.GGNN(); g.set_base(base); g.build(24, 0.5)
    indices, dists = g.query(query, 10, 0.64, 400)

Inputs may be numpy arrays or torch tensors (CPU or CUDA), C-contiguous 2-D float32/uint8;
results are torch tensors like the reference's (nb::pytorch ndarrays), int32 ids and float32
SQUARED L2 (or |1-cos|) distances.
"""
import ctypes as C
import enum
import os

import numpy as np
import torch

from . import _lib
from ._lib import check, lib


class DistanceMeasure(enum.IntEnum):
    """include/ggnn/base/def.h:27-30"""
    Euclidean = 0
    Cosine = 1


def set_log_level(level: int) -> None:
    """nanobind.cu:151"""
    lib().ggnn_set_log_level(int(level))


_NP_OF = {torch.float32: np.float32, torch.uint8: np.uint8, torch.int32: np.int32}


def _as_tensor(data, dtype=None, what="data"):
    """ndarray_to_dataset (nanobind.cu:102-110): 2-D, C-contiguous, CPU or CUDA."""
    if isinstance(data, _Dataset):
        data = data._t
    if isinstance(data, np.ndarray):
        data = torch.from_numpy(np.ascontiguousarray(data))
    if not isinstance(data, torch.Tensor):
        raise TypeError(f"{what} must be a numpy array, a torch tensor or a ggnn dataset")
    if data.dim() != 2:
        raise TypeError(f"{what} must be 2-dimensional")
    if dtype is not None and data.dtype != dtype:
        raise TypeError(f"{what} must have dtype {dtype}")
    if not data.is_contiguous():
        raise TypeError(f"{what} must be C-contiguous")
    return data


def _loc(t):
    if t.is_cuda:
        torch.cuda.current_stream(t.device).synchronize()
        return _lib.GPU, t.device.index if t.device.index is not None else 0
    return _lib.CPU, 0


def _dtype_code(t):
    if t.dtype == torch.float32:
        return _lib.F32
    if t.dtype == torch.uint8:
        return _lib.U8
    raise TypeError("unsupported datatype (float32 and uint8 are supported)")


# ---------------------------------------------------------------------------------------------
# Datasets (nanobind.cu:153-181; Dataset<T>::load/store, src/ggnn/base/dataset.cu:118-233)
# ---------------------------------------------------------------------------------------------
class _Dataset:
    _torch_dtype = None
    _suffix = None

    def __init__(self, data):
        self._t = _as_tensor(data, self._torch_dtype).clone()

    @classmethod
    def _wrap(cls, t):
        obj = cls.__new__(cls)
        obj._t = t
        return obj

    @classmethod
    def load(cls, file, from_=0, num=2 ** 32 - 1, pin_memory=False, **kw):
        """XVECS files: per vector a uint32 dimension followed by D values."""
        from_ = kw.get("from", from_)
        npdt = np.dtype(_NP_OF[cls._torch_dtype])
        with open(file, "rb") as f:
            head = np.fromfile(f, dtype=np.uint32, count=1)
            if head.size != 1:
                raise RuntimeError(f"cannot read {file}")
            D = int(head[0])
            rec = 4 + D * npdt.itemsize
            total = os.path.getsize(file) // rec
            n = max(0, min(int(num), total - int(from_)))
            f.seek(int(from_) * rec)
            raw = np.fromfile(f, dtype=np.uint8, count=n * rec).reshape(n, rec)
        data = np.ascontiguousarray(raw[:, 4:]).view(npdt).reshape(n, D)
        t = torch.from_numpy(data.copy())
        if pin_memory and torch.cuda.is_available():
            t = t.pin_memory()
        return cls._wrap(t)

    def store(self, file):
        a = self._t.cpu().numpy()
        n, D = a.shape
        rec = np.empty((n, 4 + D * a.dtype.itemsize), np.uint8)
        rec[:, :4] = np.frombuffer(np.uint32(D).tobytes(), np.uint8)
        rec[:, 4:] = a.view(np.uint8).reshape(n, -1)
        rec.tofile(file)

    @property
    def N(self):
        return int(self._t.shape[0])

    @property
    def D(self):
        return int(self._t.shape[1])

    def numel(self):
        return int(self._t.numel())

    def clone(self):
        return self._t.clone()

    @property
    def view(self):
        return self._t

    @property
    def device(self):
        return f"cuda:{self._t.device.index}" if self._t.is_cuda else "cpu"


class FloatDataset(_Dataset):
    _torch_dtype = torch.float32


class UCharDataset(_Dataset):
    _torch_dtype = torch.uint8


class IntDataset(_Dataset):
    _torch_dtype = torch.int32


class Graph:
    """Graph views (include/ggnn/base/graph.h:38-71, nanobind.cu:295-300); copies on the host."""

    def __init__(self, graph, selection, translation, nn1_stats, config):
        self.graph = graph
        self.selection = selection
        self.translation = translation
        self.nn1_stats = nn1_stats
        self.config = config


# ---------------------------------------------------------------------------------------------
# GGNN (nanobind.cu:184-268 over ggnn.cuh:42-182)
# ---------------------------------------------------------------------------------------------
_ASYNC_SLOTS = 4   # DeviceCtx::kShardStreams: slots that map to the same engine stream
_MAX_TICKETS_PER_SLOT = 64   # query_async batches whose tensors are held per slot before it is drained


class QueryTicket:
    """Result of `GGNN.query_async`: `ids, dists = ticket` works as before; `query` keeps the
    input alive; `done` is set by `synchronize()`."""
    __slots__ = ("query", "ids", "dists", "slot", "done")

    def __init__(self, query, ids, dists, slot):
        self.query, self.ids, self.dists, self.slot, self.done = query, ids, dists, slot, False

    def __iter__(self):
        return iter((self.ids, self.dists))

    def __getitem__(self, i):
        return (self.ids, self.dists)[i]

    def __len__(self):
        return 2


class GGNN:
    """GGNN main class. Provides functionality for building, loading, storing, and querying
    nearest neighbor graphs on the GPU."""

    def __init__(self):
        h = C.c_void_p()
        check(lib().ggnn_create(C.byref(h)))
        self._h = h
        self._destroy = lib().ggnn_destroy
        self._keepalive = None
        self._return_results_on_gpu = False
        self._shards = 1
        self._inflight = {}   # slot -> QueryTickets whose kernels may still be running
        self._num_gpus = 1
        self._collect_counters = False

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._destroy(h)

    def _check(self, status):
        check(status, self._h)

    def set_base(self, base):
        t = _as_tensor(base, what="base")
        loc, dev = _loc(t)
        # the binding clones its argument (nanobind.cu:102-110): the engine takes a copy
        self._check(lib().ggnn_set_base(self._h, t.data_ptr(), t.shape[0], t.shape[1],
                                        _dtype_code(t), loc, dev, 1))
        self._base_shape = tuple(t.shape)

    def set_base_reference(self, base):
        """GGNN::setBaseReference (ggnn.cuh:116-123): borrow; the tensor is kept alive here."""
        t = _as_tensor(base, what="base")
        loc, dev = _loc(t)
        self._check(lib().ggnn_set_base(self._h, t.data_ptr(), t.shape[0], t.shape[1],
                                        _dtype_code(t), loc, dev, 0))
        self._keepalive = t
        self._base_shape = tuple(t.shape)

    def set_working_directory(self, dir):
        self._check(lib().ggnn_set_working_directory(self._h, os.fspath(dir).encode()))

    def set_cpu_memory_limit(self, memory_limit):
        self._check(lib().ggnn_set_cpu_memory_limit(self._h, int(memory_limit)))

    def set_reserved_gpu_memory(self, reserved_memory):
        self._check(lib().ggnn_set_reserved_gpu_memory(self._h, int(reserved_memory)))

    def set_gpus(self, gpu_ids):
        ids = [int(g) for g in gpu_ids]
        arr = (C.c_int * len(ids))(*ids)
        self._check(lib().ggnn_set_gpus(self._h, arr, len(ids)))
        self._num_gpus = max(1, len(ids))

    def set_shard_size(self, n_shard):
        self._check(lib().ggnn_set_shard_size(self._h, int(n_shard)))
        self._n_shard = int(n_shard)

    def set_return_results_on_gpu(self, return_results_on_gpu=True):
        self._check(lib().ggnn_set_return_results_on_gpu(self._h, int(bool(return_results_on_gpu))))
        self._return_results_on_gpu = bool(return_results_on_gpu)

    def build(self, k_build, tau_build, refinement_iterations=2,
              measure=DistanceMeasure.Euclidean):
        """Build a GGNN graph."""
        self._check(lib().ggnn_build(self._h, int(k_build), float(tau_build),
                                     int(refinement_iterations), int(measure)))
        self._update_shards()

    def load(self, k_build):
        """Load a GGNN graph."""
        self._check(lib().ggnn_load(self._h, int(k_build)))
        self._update_shards()

    def store(self):
        """Store a GGNN graph."""
        self._check(lib().ggnn_store(self._h))

    def _update_shards(self):
        # shards per GPU as laid out by the engine: width factor of results kept on the GPU
        per_gpu = C.c_uint32(1)
        self._check(lib().ggnn_get_shard_layout(self._h, None, C.byref(per_gpu), None))
        self._shards = int(per_gpu.value)

    def _out(self, Nq, width, on_gpu, device):
        dev = device if on_gpu else "cpu"
        ids = torch.empty((Nq, width), dtype=torch.int32, device=dev)
        dists = torch.empty((Nq, width), dtype=torch.float32, device=dev)
        if on_gpu:
            # the blocks come from torch's caching allocator, which orders their reuse on torch's
            # stream; the engine writes them on its own stream, so work torch has queued on a
            # recycled block must have finished before the engine touches it
            torch.cuda.current_stream(device).synchronize()
        return ids, dists

    def _result_device(self, t):
        if t.is_cuda:
            return t.device
        view = _lib.GraphView()
        if lib().ggnn_get_graph(self._h, 0, C.byref(view)) == _lib.OK:
            return torch.device("cuda", view.gpu_id)
        return torch.device("cuda", torch.cuda.current_device())

    def query(self, query, k_query, tau_query, max_iterations=400,
              measure=DistanceMeasure.Euclidean):
        """Run a query and return indices and distances."""
        t = _as_tensor(query, what="query")
        loc, dev = _loc(t)
        on_gpu = self._return_results_on_gpu
        width = int(k_query) * (self._shards if on_gpu else 1)
        ids, dists = self._out(t.shape[0], width, on_gpu, self._result_device(t) if on_gpu else None)
        self._check(lib().ggnn_query(self._h, t.data_ptr(), t.shape[0], t.shape[1],
                                     _dtype_code(t), loc, dev, int(k_query), float(tau_query),
                                     int(max_iterations), int(measure), ids.data_ptr(),
                                     dists.data_ptr(), _lib.GPU if on_gpu else _lib.CPU))
        return ids, dists

    def query_async(self, query, k_query, tau_query, max_iterations=400,
                    measure=DistanceMeasure.Euclidean, slot=0):
        """Extension for serving: enqueue a batch and return a `QueryTicket` (unpacks like the
        `(ids, dists)` pair) whose GPU tensors are valid after `synchronize()`.  Batches with
        different `slot`s overlap on the device (one GPU, query on that GPU); the result has the
        results-on-GPU shape [Nq, k_query * shards].

        Lifetime: the kernels run on the engine's own streams, which torch's caching allocator
        knows nothing about.  The engine object therefore keeps the query tensor and both
        result tensors referenced until `synchronize()` (of that slot) has returned, whatever
        the caller does with its own references -- a temporary passed as `query`, or a
        rebound loop variable, cannot be recycled under a running kernel.  At most
        `_MAX_TICKETS_PER_SLOT` batches are held per slot: enqueueing one more first synchronises
        that slot and releases them."""
        t = _as_tensor(query, what="query")
        if self._num_gpus > 1 or _lib.get_hook("EXCHANGE") == 1:
            # several GPUs (or the forced RCCL path of the tests): merged [Nq, k] results; host-side tensors are page-locked so that the
            # engine's copies stay asynchronous
            if not t.is_cuda and not t.is_pinned():
                t = t.pin_memory()
            dev = t.device.index if t.is_cuda else -1
            if t.is_cuda:
                ids, dists = self._out(t.shape[0], int(k_query), True, t.device)
            else:
                ids = torch.empty((t.shape[0], int(k_query)), dtype=torch.int32, pin_memory=True)
                dists = torch.empty((t.shape[0], int(k_query)), dtype=torch.float32,
                                    pin_memory=True)
        else:
            if not t.is_cuda:
                raise RuntimeError("query_async needs the query on the GPU")
            loc, dev = _loc(t)
            ids, dists = self._out(t.shape[0], int(k_query) * self._shards, True, t.device)
        self._check(lib().ggnn_query_async(self._h, t.data_ptr(), t.shape[0], t.shape[1],
                                           _dtype_code(t), dev, int(k_query), float(tau_query),
                                           int(max_iterations), int(measure), ids.data_ptr(),
                                           dists.data_ptr(), int(slot)))
        ticket = QueryTicket(t, ids, dists, int(slot))
        held = self._inflight.setdefault(int(slot) % _ASYNC_SLOTS, [])
        if len(held) >= _MAX_TICKETS_PER_SLOT:
            # a serving loop that waits some other way (its own events, torch.cuda.synchronize)
            # never calls synchronize(): the references held for the kernels' sake must not grow
            # without bound -- the slot is drained (its batches ran long ago) and released
            self.synchronize(slot)
            held = self._inflight.setdefault(int(slot) % _ASYNC_SLOTS, [])
        held.append(ticket)
        return ticket

    def synchronize(self, slot=None):
        """wait for every batch enqueued with query_async (or only for those of one slot); the
        tensors of the finished batches are released to their owners"""
        if slot is None:
            self._check(lib().ggnn_synchronize(self._h))
            done = [t for ts in self._inflight.values() for t in ts]
            self._inflight.clear()
        else:
            self._check(lib().ggnn_synchronize_slot(self._h, int(slot)))
            done = self._inflight.pop(int(slot) % _ASYNC_SLOTS, [])
        for t in done:
            t.done = True

    def bf_query(self, query, k_gt=100, measure=DistanceMeasure.Euclidean):
        """Run a brute-force query and indices and distances."""
        t = _as_tensor(query, what="query")
        loc, dev = _loc(t)
        on_gpu = self._return_results_on_gpu
        ids, dists = self._out(t.shape[0], int(k_gt), on_gpu,
                               self._result_device(t) if on_gpu else None)
        self._check(lib().ggnn_bf_query(self._h, t.data_ptr(), t.shape[0], t.shape[1],
                                        _dtype_code(t), loc, dev, int(k_gt), int(measure),
                                        ids.data_ptr(), dists.data_ptr(),
                                        _lib.GPU if on_gpu else _lib.CPU))
        return ids, dists

    def get_graph(self, on_gpu_shard_id=0):
        """Access the GGNN graph."""
        view = _lib.GraphView()
        self._check(lib().ggnn_get_graph(self._h, int(on_gpu_shard_id), C.byref(view)))
        cfg = view.config
        K = cfg.KBuild

        def fetch(ptr, count, dtype):
            if count == 0:
                return torch.empty(0, dtype=dtype)
            out = torch.empty(count, dtype=dtype)
            nbytes = count * out.element_size()
            with torch.cuda.device(view.gpu_id):
                tmp = torch.empty(count, dtype=dtype, device="cuda")
                _hip_memcpy_d2d(tmp.data_ptr(), ptr, nbytes)
                out.copy_(tmp)
            return out

        graph_all = fetch(view.graph, cfg.N_all * K, torch.int32).view(cfg.N_all, K)
        tr_all = fetch(view.translation, cfg.ST_all, torch.int32)
        sel_all = fetch(view.selection, cfg.ST_all, torch.int32)
        stats = fetch(view.nn1_stats, 2, torch.float32).view(2, 1)
        graph, selection, translation = [], [], []
        for l in range(4):
            g = graph_all[cfg.Ns_offsets[l]:cfg.Ns_offsets[l] + cfg.Ns[l]]
            graph.append(IntDataset._wrap(g))
            if l:
                s = slice(cfg.STs_offsets[l], cfg.STs_offsets[l] + cfg.Ns[l])
                selection.append(IntDataset._wrap(sel_all[s].view(-1, 1)))
                translation.append(IntDataset._wrap(tr_all[s].view(-1, 1)))
            else:
                selection.append(IntDataset._wrap(torch.empty((0, 1), dtype=torch.int32)))
                translation.append(IntDataset._wrap(torch.empty((0, 1), dtype=torch.int32)))
        return Graph(graph, selection, translation, FloatDataset._wrap(stats), cfg.as_dict())

    # tracing helpers (not part of the reference surface)
    def last_timing_ms(self):
        b, q, f = C.c_float(), C.c_float(), C.c_float()
        self._check(lib().ggnn_last_timing_ms(self._h, C.byref(b), C.byref(q), C.byref(f)))
        return {"build_ms": b.value, "query_ms": q.value, "bf_query_ms": f.value}

    def set_collect_counters(self, enable=True):
        self._check(lib().ggnn_set_collect_counters(self._h, int(bool(enable))))
        self._collect_counters = bool(enable)

    def last_query_parts(self):
        """half-batches the last blocking multi-GPU query() was searched in (2: the second half's
        search overlapped the exchange and merge of the first)"""
        n = C.c_uint32()
        self._check(lib().ggnn_last_query_parts(self._h, C.byref(n)))
        return int(n.value)

    def rccl_ranks(self):
        """ranks of the RCCL communicator behind the handle's exchange (0: none in use)"""
        n = C.c_uint32()
        self._check(lib().ggnn_rccl_ranks(self._h, C.byref(n)))
        return int(n.value)

    def last_build_work(self):
        """work counters and kernel times of the merge / sym launches of the last build()
        (set_collect_counters(True) before the build): {"merge": {...}, "sym": {...}}"""
        w = _lib.BuildWork()
        self._check(lib().ggnn_last_build_work(self._h, C.byref(w)))
        return {"merge": w.merge.as_dict(), "sym": w.sym.as_dict()}

    def set_prescreen(self, enable=True):
        """Exact pre-screen of float32/Euclidean queries on an 8-bit copy of the base (an
        extension: same results, less memory traffic; costs N x D bytes per shard)."""
        self._check(lib().ggnn_set_prescreen(self._h, int(bool(enable))))

    def set_build_hooks(self, rng=None, serial_sym=False):
        """Deterministic build (see ggnn_set_build_hooks): `rng` = [3, N_shard] float32 uniform
        (0, 1] numbers for the selection kernel, `serial_sym` = sym one point per launch in
        ascending order.  Used to compare a whole build with a CPU restatement bit for bit."""
        if rng is None:
            self._check(lib().ggnn_set_build_hooks(self._h, None, 0, int(bool(serial_sym))))
            return
        arr = np.ascontiguousarray(np.asarray(rng, dtype=np.float32))
        self._check(lib().ggnn_set_build_hooks(self._h, arr.ctypes.data, arr.size,
                                               int(bool(serial_sym))))

    def last_exchange(self):
        """how the last query combined per-GPU results: "none", "rccl" or "copy" """
        return lib().ggnn_last_exchange(self._h).decode()

    def last_bf_query_rescanned(self):
        """queries of the last bf_query answered by the exhaustive scan because the matrix-core
        pre-selection could not be certified exact (results are exact either way)"""
        n = C.c_uint32()
        self._check(lib().ggnn_last_bf_query_rescanned(self._h, C.byref(n)))
        return int(n.value)

    def last_query_counters(self):
        d, p = C.c_uint64(), C.c_uint64()
        self._check(lib().ggnn_last_query_counters(self._h, C.byref(d), C.byref(p)))
        return {"n_dist": d.value, "n_pop": p.value}

    def last_query_rows_read(self):
        """float rows and 8-bit pre-screen rows read by the last query (with collect_counters)"""
        f, c = C.c_uint64(), C.c_uint64()
        self._check(lib().ggnn_last_query_rows_read(self._h, C.byref(f), C.byref(c)))
        return {"float_rows": f.value, "code_rows": c.value}


_hip = None


def _hip_memcpy_d2d(dst, src, nbytes):
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemcpy.restype = C.c_int
    rc = _hip.hipMemcpy(dst, src, nbytes, 3)  # hipMemcpyDeviceToDevice
    if rc != 0:
        raise RuntimeError(f"hipMemcpy failed with {rc}")


# ---------------------------------------------------------------------------------------------
# Evaluator (include/ggnn/base/eval.h:31-65, src/ggnn/base/eval.cpp:37-242)
# ---------------------------------------------------------------------------------------------
class Evaluation:
    def __init__(self, k_query, c1, c1_dup, c_k_query, c_k_query_dup, r_k_query, r_k_query_dup):
        self.k_query = k_query
        self.c1 = c1
        self.c1_dup = c1_dup
        self.c_k_query = c_k_query
        self.c_k_query_dup = c_k_query_dup
        self.r_k_query = r_k_query
        self.r_k_query_dup = r_k_query_dup

    def __repr__(self):
        # operator<<, eval.cpp:67-86
        def dup(v, nl):
            if not np.isnan(v):
                return f" +duplicates: {v:g}" + ("\n" if nl else "")
            return " (duplicates unknown)" + ("\n" if nl else "")
        return (f"c@1 (=r@1): {self.c1:g}" + dup(self.c1_dup, True) +
                f"c@{self.k_query}: {self.c_k_query:g}" + dup(self.c_k_query_dup, True) +
                f"r@{self.k_query}: {self.r_k_query:g}" + dup(self.r_k_query_dup, False))


def _eval_distance(base_rows, query_rows, measure):
    """compute_distance, eval.cpp:37-65 incl. its quirks (Euclidean WITH sqrt; the cosine
    variant uses the base vector for both norms)."""
    a = base_rows.astype(np.float32)
    b = query_rows.astype(np.float32)
    if measure == DistanceMeasure.Euclidean:
        return np.sqrt(((a - b) ** 2).sum(-1, dtype=np.float32))
    dot = (a * b).sum(-1, dtype=np.float32)
    na = (a * a).sum(-1, dtype=np.float32)
    prod = na * na
    with np.errstate(divide="ignore", invalid="ignore"):
        d = np.abs(np.float32(1.0) - dot / np.sqrt(prod))
    return np.where(prod > 0, d, np.float32(1.0)).astype(np.float32)


class Evaluator:
    def __init__(self, base, query, gt, k_query, measure=DistanceMeasure.Euclidean):
        self.k_query = int(k_query)
        self.measure = DistanceMeasure(measure)
        gt_t = _as_tensor(gt, torch.int32, "gt")
        if gt_t.is_cuda:
            raise RuntimeError("Ground truth data needs to be given on the CPU for evaluation.")
        self.gt = gt_t.numpy().copy()
        self.top1_end = None
        self.topk_end = None
        base_t = _as_tensor(base, what="base")
        query_t = _as_tensor(query, what="query")
        if base_t.shape[0] == 0 or query_t.shape[0] == 0 or base_t.is_cuda or query_t.is_cuda:
            return  # duplicates unknown (eval.cpp:93-102)
        if base_t.dtype != query_t.dtype:
            raise RuntimeError("base and query need to have the same data type")
        b, q = base_t.numpy(), query_t.numpy()
        Nq, gtD, K = q.shape[0], self.gt.shape[1], self.k_query
        eps = np.float32(0.000001)
        # distances of all ground-truth entries, [Nq, gtD]
        gd = np.stack([_eval_distance(b[self.gt[:, k]], q, self.measure) for k in range(gtD)], 1)

        def run_length(ref, start):
            # number of consecutive entries from `start` whose distance stays within eps
            within = (gd[:, start:] - ref[:, None]) <= eps
            if within.shape[1] == 0:
                return np.zeros(Nq, np.uint32)
            stop = np.where(within.all(1), within.shape[1], np.argmin(within, 1))
            return stop.astype(np.uint32)

        self.top1_end = 1 + run_length(gd[:, 0], 1)
        if K <= gtD:
            self.topk_end = K + run_length(gd[:, K - 1], K)
        else:
            self.topk_end = np.full(Nq, gtD, np.uint32)

    def evaluate_results(self, results):
        """Evaluate the accuracy of a query result."""
        res_t = _as_tensor(results, torch.int32, "results")
        if res_t.is_cuda:
            raise RuntimeError("Results need to be given on the CPU for evaluation.")
        res = res_t.numpy()
        K, gt = self.k_query, self.gt
        if gt.shape[1] == 0:
            raise RuntimeError("No ground truth data loaded. cannot compute accuracy.")
        N = res.shape[0]
        has_dup = self.top1_end is not None
        end1 = self.top1_end[:N] if has_dup else np.ones(N, np.uint32)
        endk = self.topk_end[:N] if has_dup else np.full(N, K, np.uint32)
        gtD = gt.shape[1]
        kg = np.arange(gtD)[None, None, :]                       # [1,1,gtD]
        match = res[:N, :K, None] == gt[:N, None, :]             # [N,K,gtD]
        match &= kg < endk[:, None, None]
        first_res = np.zeros((1, K, 1), bool)
        first_res[0, 0, 0] = True
        c1 = int((match[:, :, :1] & first_res).sum())
        r_k_dup = int(match[:, :, 0].sum())
        r_k = r_k_dup if K > 0 else 0
        c1_dup = int((match & first_res & (kg < end1[:, None, None])).sum())
        c_k = int((match & (kg < K)).sum())
        c_k_dup = int(match.sum())
        inv_q = np.float32(1.0) / np.float32(N)
        inv_r = np.float32(1.0) / np.float32(N * K)
        nan = float("nan")
        return Evaluation(K, float(c1 * inv_q), float(c1_dup * inv_q) if has_dup else nan,
                          float(c_k * inv_r), float(c_k_dup * inv_r) if has_dup else nan,
                          float(r_k * inv_q), float(r_k_dup * inv_q) if has_dup else nan)
