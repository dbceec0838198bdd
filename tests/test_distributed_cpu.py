"""world_size-2 gloo test of the base-sharded multi-GPU path (ggnn_amd.distributed) on CPU: the
rank-local engine is replaced by an oracle-backed stand-in, everything else (partitioning, id
offsets, all-gather, k-way merge) is the product code.  The merged result must equal the
reference's ResultMerger semantics (oracle.merge_results) and, for brute force, the exact
global answer."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class OracleEngine:
    """CPU stand-in with the GGNN surface, backed by the oracle (test infrastructure)."""

    def __init__(self):
        from oracle import oracle
        self.orc = oracle

    def set_base(self, base):
        self.base = base.numpy().copy()

    def build(self, k_build, tau_build, refinement_iterations=2, measure=0):
        N = self.base.shape[0]
        self.cfg, self.graph, self.tr, self.sel, self.stats = self.orc.build(
            self.base, k_build, tau_build, refinement_iterations, int(measure),
            rng=self.orc.make_rng(N, 7), threads=2)

    def query(self, query, k, tau, iters=400, measure=0):
        c = self.cfg
        start = self.tr[c.STs_offsets[3]:c.STs_offsets[3] + c.Ns[3]]
        ids, d = self.orc.query(self.base, query.numpy(), self.graph[:c.N], start, self.stats, k,
                                tau, iters, int(measure), threads=2)
        return torch.from_numpy(ids), torch.from_numpy(d)

    def bf_query(self, query, k, measure=0):
        ids, d = self.orc.bf_query(self.base, query.numpy(), k, int(measure), threads=2)
        return torch.from_numpy(ids), torch.from_numpy(d)

    # the asynchronous surface (nothing is asynchronous on the CPU: the ticket holds the result)
    def query_async(self, query, k, tau, iters=400, measure=0, slot=0):
        self.slots_used = getattr(self, "slots_used", []) + [slot]
        return self.query(query, k, tau, iters, measure)

    def synchronize(self, slot=None):
        pass


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import make_int_data
        from ggnn_amd.distributed import ShardedGGNN, partition_bounds
        from oracle import oracle as orc
        N, D, K = 1200, 32, 10
        base = torch.from_numpy(make_int_data(N, D, 5))
        query = torch.from_numpy(make_int_data(40, D, 6))
        sg = ShardedGGNN(engine=OracleEngine())
        sg.set_base(base)
        lo, hi = partition_bounds(N, world, rank)
        assert (lo, hi) == (rank * N // world, (rank + 1) * N // world) and sg.n_local == hi - lo
        sg.build(24, 0.5, 0)
        # brute force through the sharded path == exact global brute force
        ids, d = sg.bf_query(query, K)
        g_ids, g_d = orc.bf_query(base.numpy(), query.numpy(), K)
        assert np.array_equal(ids.numpy(), g_ids) and np.array_equal(d.numpy(), g_d)
        # graph query: merged result == ResultMerger semantics over the per-rank results
        ids, d = sg.query(query, K, 0.6, 200)
        l_ids, l_d = sg.engine.query(query, K, 0.6, 200)
        parts_i = [torch.empty_like(l_ids) for _ in range(world)]
        parts_d = [torch.empty_like(l_d) for _ in range(world)]
        dist.all_gather(parts_i, l_ids)
        dist.all_gather(parts_d, l_d)
        r_ids, r_d = orc.merge_results([p.numpy() for p in parts_i], [p.numpy() for p in parts_d],
                                       K, 1, N // world)
        # distances always agree; ids agree wherever the distance is unique (the reference's heap
        # order on exact ties is implementation-defined)
        assert np.array_equal(d.numpy(), r_d)
        uniq = np.ones_like(r_ids, bool)
        uniq[:, 1:] &= r_d[:, 1:] != r_d[:, :-1]
        uniq[:, :-1] &= r_d[:, :-1] != r_d[:, 1:]
        assert np.array_equal(ids.numpy()[uniq], r_ids[uniq])
        assert ids.numpy().min() >= 0 and ids.numpy().max() < N
        # a blocking batch split into two half-batches in flight (default from 4096 queries): the
        # same collective sequence on every rank, the concatenated result identical
        assert sg.last_query_parts == 1
        sg.split_blocking = True
        ids2, d2 = sg.query(query, K, 0.6, 200)
        assert sg.last_query_parts == 2 and sg.engine.slots_used[-2:] == [0, 1]
        assert torch.equal(ids2, ids) and torch.equal(d2, d)
        ids3, d3 = sg.query(query[:7], K, 0.6, 200)       # odd count: halves of 3 and 4
        assert torch.equal(ids3, ids[:7]) and torch.equal(d3, d[:7])
        # An engine whose asynchronous lanes refuse the batch (the real one does for shards that
        # take turns on the GPU: GGNN_UNSUPPORTED) ON ONE RANK ONLY -- swapping depends on the
        # rank's free memory: the ranks agree on the path before any data collective, all fall back
        # to ONE blocking search + one exchange, same result (round-5 advisor finding: a rank that
        # fell back alone would issue one all-gather of nq rows against its peers' two of nq / 2)
        from ggnn_amd._lib import GGNNError, UNSUPPORTED
        real_async = sg.engine.query_async

        def refusing(*a, **kw):
            raise GGNNError(UNSUPPORTED, "query_async: not available while shards are swapped")
        if rank == world - 1:
            sg.engine.query_async = refusing
        ids4, d4 = sg.query(query, K, 0.6, 200)
        assert sg.last_query_parts == 1
        assert torch.equal(ids4, ids) and torch.equal(d4, d)
        # any OTHER failure on one rank (out of memory, device error ...) is an error on every
        # rank, raised before a collective is entered -- not a silent fallback, not a hang

        def failing(*a, **kw):
            raise MemoryError("out of memory allocating the result buffers")
        sg.engine.query_async = failing if rank == 0 else real_async
        try:
            sg.query(query, K, 0.6, 200)
            raise AssertionError("a failure on one rank must surface on every rank")
        except MemoryError:
            assert rank == 0
        except RuntimeError as e:
            assert rank != 0 and "another rank failed" in str(e)
        # ... and the group is still usable afterwards
        sg.engine.query_async = real_async
        ids5, d5 = sg.query(query, K, 0.6, 200)
        assert sg.last_query_parts == 2 and torch.equal(ids5, ids) and torch.equal(d5, d)
        sg.engine.query_async = real_async
        sg.split_blocking = None
        with open(os.path.join(tmp, f"ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def test_sharded_query_gloo_world2(tmp_path, orc):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


class OracleShardsEngine(OracleEngine):
    """the stand-in with several shards per rank (GGNN::setShardSize on the rank's slice): one
    independent graph per shard, the local result is the reference's per-GPU row -- the K results
    of every shard with ids offset by on_gpu_shard * N_shard, sorted (gpu_instance.cu:745-790)"""

    def set_shard_size(self, n_shard):
        self.n_shard = int(n_shard)

    def build(self, k_build, tau_build, refinement_iterations=2, measure=0):
        n = self.n_shard
        self.shards = []
        for lo in range(0, self.base.shape[0], n):
            b = self.base[lo:lo + n]
            self.shards.append((b,) + tuple(self.orc.build(
                b, k_build, tau_build, refinement_iterations, int(measure),
                rng=self.orc.make_rng(n, 7), threads=1)))

    def query(self, query, k, tau, iters=400, measure=0):
        q = query.numpy()
        ids, ds = [], []
        for s, (b, c, graph, tr, sel, stats) in enumerate(self.shards):
            start = tr[c.STs_offsets[3]:c.STs_offsets[3] + c.Ns[3]]
            i, d = self.orc.query(b, q, graph[:c.N], start, stats, k, tau, iters, int(measure),
                                  threads=1)
            ids.append(i + s * self.n_shard)
            ds.append(d)
        i, d = self.orc.sort_shard_results(np.concatenate(ids, 1), np.concatenate(ds, 1))
        return torch.from_numpy(i), torch.from_numpy(d)


def _worker8(rank, world, port, tmp):
    """what an 8-GPU node runs first: 8 ranks, two shards per rank, a query count that 8 does not
    divide, blocking (split into two half-batches of different sizes) and batches in flight"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import make_int_data
        from ggnn_amd.distributed import ShardedGGNN
        from oracle import oracle as orc
        N, D, K, NSH = 3200, 16, 10, 200            # 8 ranks x 2 shards x 200 points
        base = torch.from_numpy(make_int_data(N, D, 15))
        query = torch.from_numpy(make_int_data(37, D, 16))   # 37 = 8 * 4 + 5
        sg = ShardedGGNN(engine=OracleShardsEngine())
        sg.set_base(base)
        assert sg.n_local == N // world
        sg.set_shard_size(NSH)
        sg.build(16, 0.5, 0)
        l_ids, l_d = sg.engine.query(query, K, 0.7, 100)
        assert l_ids.shape == (37, 2 * K)
        parts_i = [torch.empty_like(l_ids) for _ in range(world)]
        parts_d = [torch.empty_like(l_d) for _ in range(world)]
        dist.all_gather(parts_i, l_ids)
        dist.all_gather(parts_d, l_d)
        # the reference's ResultMerger over the 8 per-GPU rows: id offset g * shards_per_gpu * N_shard
        r_ids, r_d = orc.merge_results([p.numpy() for p in parts_i], [p.numpy() for p in parts_d],
                                       K, 2, NSH)
        for split in (False, True):
            sg.split_blocking = split
            ids, d = sg.query(query, K, 0.7, 100)
            assert sg.last_query_parts == (2 if split else 1)
            assert np.array_equal(d.numpy(), r_d), split
            uniq = np.ones_like(r_ids, bool)
            uniq[:, 1:] &= r_d[:, 1:] != r_d[:, :-1]
            uniq[:, :-1] &= r_d[:, :-1] != r_d[:, 1:]
            assert np.array_equal(ids.numpy()[uniq], r_ids[uniq]), split
            assert ids.numpy().min() >= 0 and ids.numpy().max() < N
        # two batches in flight, finished in order
        t0 = sg.query_async(query[:20], K, 0.7, 100, slot=0)
        t1 = sg.query_async(query[20:], K, 0.7, 100, slot=1)
        a, b = sg.finish(t0), sg.finish(t1)
        assert np.array_equal(torch.cat([a[1], b[1]]).numpy(), r_d)
        # exact brute force through the sharded path
        ids, d = sg.bf_query(query, K)
        g_ids, g_d = orc.bf_query(base.numpy(), query.numpy(), K)
        assert np.array_equal(d.numpy(), g_d)
        with open(os.path.join(tmp, f"ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def test_sharded_query_gloo_world8_two_shards_per_rank(tmp_path, orc):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker8, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(8))


def test_merge_gathered_cpu_matches_oracle(orc):
    from ggnn_amd.distributed import merge_gathered
    r = np.random.default_rng(3)
    P, Nq, K = 4, 50, 10
    d = np.sort(r.permutation(P * Nq * K * 2)[:P * Nq * K].astype(np.float32).reshape(P, Nq, K), 2)
    ids = r.integers(0, 500, (P, Nq, K)).astype(np.int32)
    m_ids, m_d = merge_gathered(torch.from_numpy(ids), torch.from_numpy(d), K, 500)
    o_ids, o_d = orc.merge_results(list(ids), list(d), K, 1, 500)
    assert np.array_equal(m_ids.numpy(), o_ids) and np.array_equal(m_d.numpy(), o_d)


def test_partition_bounds():
    from ggnn_amd.distributed import partition_bounds
    assert partition_bounds(100, 4, 3) == (75, 100)
    with pytest.raises(RuntimeError):
        partition_bounds(10, 3, 0)
