"""Base-sharded multi-GPU search: one process per GPU, RCCL all-gather of the candidates.

Reference behaviour being replaced (src/ggnn/base/ggnn.cu:278-330, result_merger.cpp:51-149):
the base is split contiguously into num_gpus x shards_per_gpu shards, one std::thread per GPU
searches the FULL query set in its shards, results are copied D2H and heap-merged on the CPU
with global id = partition * shards_per_gpu * N_shard + key.

MI355X-native form: every rank owns one GGNN engine over its slice of the base.  A query runs
the local traversal, keeps the [Nq, K] candidates on the device, exchanges the packed
(ids, dists) of all ranks with ONE all-gather each over xGMI (Nq*K*8 bytes per rank -- latency
bound, so a single un-bucketed collective is right) and merges on the device
(ggnn_op_merge_results).  With backend "gloo" the same code runs on CPU tensors for tests; the
merge then uses the device-independent torch implementation below.
"""
import torch
import torch.distributed as dist

from ._lib import UNSUPPORTED
from .api import GGNN, DistanceMeasure, _as_tensor


def partition_bounds(N, world_size, rank):
    """contiguous equal split (ggnn.cu:160-200); N must be divisible by world_size"""
    if N % world_size:
        raise RuntimeError("base.N needs to be evenly divisible by (N_shard x num_gpus).")
    n = N // world_size
    return rank * n, (rank + 1) * n


def merge_gathered(parts_ids, parts_dists, k, id_offset_per_part):
    """k-way merge of [P, Nq, stride] sorted rows; ties -> lower part first.  torch
    implementation used for CPU tensors (gloo tests); CUDA tensors use the HIP kernel."""
    if parts_ids.is_cuda:
        from . import ops
        return ops.merge_results(parts_ids.contiguous(), parts_dists.contiguous(), k,
                                 id_offset_per_part)
    P, Nq, stride = parts_ids.shape
    offs = (torch.arange(P, dtype=torch.int32) * id_offset_per_part).view(P, 1, 1)
    ids = (parts_ids + offs).permute(1, 0, 2).reshape(Nq, P * stride)
    dists = parts_dists.permute(1, 0, 2).reshape(Nq, P * stride)
    order = torch.sort(dists, dim=1, stable=True).indices[:, :k]
    return torch.gather(ids, 1, order).contiguous(), torch.gather(dists, 1, order).contiguous()


class ShardedGGNN:
    """`GGNN` over a base that is partitioned across the ranks of a process group.

    Every rank calls the same methods with the same arguments; `set_base` takes either the whole
    base (each rank keeps its slice) or, with `is_local_slice=True`, the rank's own slice.
    """

    def __init__(self, group=None, engine=None):
        """engine: object with the GGNN surface (tests inject a CPU stand-in); default: a GGNN
        engine on the current device that keeps its results on the GPU."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed needs to be initialised first")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        if engine is None:
            engine = GGNN()
            engine.set_return_results_on_gpu(True)
        self.engine = engine
        self.n_local = None
        # blocking query() as two half-batches in flight: None = from 4096 queries, True / False
        self.split_blocking = None
        self.last_query_parts = 1

    def set_base(self, base, is_local_slice=False):
        t = _as_tensor(base, what="base")
        if not is_local_slice:
            lo, hi = partition_bounds(t.shape[0], self.world_size, self.rank)
            t = t[lo:hi].contiguous()
        self.n_local = int(t.shape[0])
        self.engine.set_base(t)

    def set_shard_size(self, n_shard):
        """several resident shards per rank (GGNN::setShardSize on the rank's slice): the local
        candidates are then sorted rows of K * shards_per_rank entries, merged across ranks with
        id offset rank * n_local as before"""
        self.engine.set_shard_size(n_shard)

    def build(self, k_build, tau_build, refinement_iterations=2,
              measure=DistanceMeasure.Euclidean):
        self.engine.build(k_build, tau_build, refinement_iterations, measure)

    def _exchange(self, ids, dists, k):
        P = self.world_size
        nq, stride = ids.shape
        out_device = ids.device
        # ONE collective: ids and the bit patterns of the distances travel in the same int32 buffer
        packed = torch.cat([ids, dists.view(torch.int32)], dim=1).contiguous()
        if packed.is_cuda and dist.get_backend(self.group) == "gloo":
            # backend without device collectives: stage the (small) candidate lists through the host
            packed = packed.cpu()
        # dim-0 concatenation layout (valid for RCCL and gloo): [P*Nq, 2*stride]
        gathered = torch.empty((P * nq, 2 * stride), dtype=torch.int32, device=packed.device)
        dist.all_gather_into_tensor(gathered, packed, group=self.group)
        gathered = gathered.to(out_device).view(P, nq, 2 * stride)
        g_ids = gathered[:, :, :stride].contiguous()
        g_dists = gathered[:, :, stride:].contiguous().view(torch.float32)
        return merge_gathered(g_ids, g_dists, k, self.n_local)

    def query(self, query, k_query, tau_query, max_iterations=400,
              measure=DistanceMeasure.Euclidean):
        """every rank returns the merged global [Nq, K] result (device tensors).

        A batch of 4096 queries or more (`split_blocking`) runs as two half-batches in flight:
        the local search of the second half is enqueued before the first is exchanged and merged,
        so the all-gather over xGMI and the merge hide behind a traversal instead of following it
        -- the pipelining of query_async / finish without the caller having to use it.  The
        result is the concatenation, bit-identical to the unsplit call."""
        t = _as_tensor(query, what="query")
        nq = int(t.shape[0])
        split = self.split_blocking if self.split_blocking is not None else nq >= 4096
        if split and nq >= 2 and self._can_split(t):
            h = nq // 2
            tickets, err, code = [], None, 0
            try:
                tickets.append(self.query_async(t[:h], k_query, tau_query, max_iterations, measure,
                                                slot=0))
                tickets.append(self.query_async(t[h:], k_query, tau_query, max_iterations, measure,
                                                slot=1))
            except Exception as e:   # noqa: BLE001 -- classified, agreed on and re-raised below
                # 1: the asynchronous lanes refuse what the blocking call accepts (an engine whose
                # shards take turns on the GPU: GGNN_UNSUPPORTED) -> fall back; 2: a real failure
                # (out of memory, device error, ...) -> raise.  Either may happen on ONE rank only
                # (swapping depends on the rank's free memory), so the ranks AGREE on the path
                # before any of them enters a data collective: a rank that fell back alone would
                # issue one all-gather of nq rows while its peers issue two of nq / 2.
                err, code = e, (1 if getattr(e, "status", None) == UNSUPPORTED else 2)
            worst = self._agree(code)
            if worst == 0:
                a = self.finish(tickets[0])
                b = self.finish(tickets[1])
                self.last_query_parts = 2
                return torch.cat([a[0], b[0]]), torch.cat([a[1], b[1]])
            for tk in tickets:          # what this rank did enqueue is drained and dropped
                self.engine.synchronize(tk[2])
            if worst >= 2:
                if code == 2:
                    raise err
                raise RuntimeError("ShardedGGNN.query: another rank failed to enqueue its local "
                                   "search; no collective was entered")
        self.last_query_parts = 1
        ids, dists = self.engine.query(t, k_query, tau_query, max_iterations, measure)
        return self._exchange(ids, dists, int(k_query))

    def _agree(self, code):
        """MAX over the ranks of a small status code (one 4-byte all-reduce; the searches that
        were enqueued run meanwhile on the engine's own streams)"""
        if self.world_size == 1:
            return code
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        f = torch.tensor([code], dtype=torch.int32, device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MAX, group=self.group)
        return int(f.item())

    def _can_split(self, t):
        """the preconditions of the engine's asynchronous lanes, as the C-level split checks them
        (engine_query.cpp query_split): the query on the GPU, rows that need no padding (16-byte
        multiples), work counters off (they belong to one blocking launch) -- anything else keeps
        the single blocking engine.query(), which accepts all of it"""
        if not hasattr(self.engine, "query_async"):
            return False
        if not isinstance(self.engine, GGNN):
            return True   # a stand-in engine (CPU tests) states its own limits by raising
        if not t.is_cuda or (t.shape[1] * t.element_size()) % 16 or not t.is_contiguous():
            return False
        if getattr(self.engine, "_collect_counters", False):
            return False
        return True

    def bf_query(self, query, k_gt=100, measure=DistanceMeasure.Euclidean):
        ids, dists = self.engine.bf_query(query, k_gt, measure)
        return self._exchange(ids, dists, int(k_gt))

    # batches in flight: the local search of batch i+1 is enqueued before batch i is exchanged, so
    # the thin tail of a rank's 10k-wave launch, the all-gather and the merge all overlap with the
    # next batch's traversal (every rank must call these in the same order)
    def query_async(self, query, k_query, tau_query, max_iterations=400,
                    measure=DistanceMeasure.Euclidean, slot=0):
        """enqueue the local search of one batch; returns a ticket for `finish`"""
        # the engine's ticket keeps query, ids and dists alive until its slot is synchronised
        local = self.engine.query_async(query, k_query, tau_query, max_iterations, measure, slot)
        return (local, int(k_query), int(slot))

    def finish(self, ticket):
        """wait for the ticket's local search, exchange and merge: the global [Nq, K] result"""
        local, k, slot = ticket
        ids, dists = local
        self.engine.synchronize(slot)
        return self._exchange(ids, dists, k)
