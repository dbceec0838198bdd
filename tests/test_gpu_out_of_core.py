"""shards that do not fit next to each other on their GPU take turns -- GPU slots <->
page-locked host buffers <-> part files (the reference's three tiers, gpu_instance.cu:135-227,
371-497).  The test box has 288 GB, so the mode is forced with the hook RESIDENT_SHARDS; the
build is made deterministic with the build hooks (selection numbers injected, sym serial) so that
the out-of-core handle and an all-resident handle hold the SAME graphs and every result can be
compared bit for bit."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N, D, K, N_SHARD = 12000, 64, 24, 3000        # four shards on one GPU


def _data(seed, n=N):
    return (np.random.default_rng(seed).integers(0, 52, (n, D)) * 5).astype(np.float32)


def _rng(orc):
    return orc.make_rng(N_SHARD, 5)[:3]


def _build(base, rng, slots, gpus=(0,), workdir=None, cpu_limit=None, on_gpu=False):
    import ggnn_amd as ggnn
    from ggnn_amd import _lib
    eng = ggnn.GGNN()
    if on_gpu:
        eng.set_base_reference(base)
    else:
        eng.set_base(torch.from_numpy(base))
    eng.set_gpus(list(gpus))
    eng.set_shard_size(N_SHARD)
    if workdir is not None:
        eng.set_working_directory(workdir)
    if cpu_limit is not None:
        eng.set_cpu_memory_limit(cpu_limit)
    eng.set_build_hooks(rng, serial_sym=True)
    with _lib.hooks(RESIDENT_SHARDS=slots):
        eng.build(K, 0.5, 1)
    return eng


def _graphs(eng, shards):
    out = []
    for s in range(shards):
        g = eng.get_graph(s)
        out.append((np.concatenate([g.graph[l].view.numpy().reshape(-1, K) for l in range(4)]),
                    np.concatenate([g.translation[l].view.numpy().reshape(-1) for l in range(1, 4)]),
                    g.nn1_stats.view.numpy().reshape(-1).copy()))
    return out


@pytest.mark.parametrize("slots", [1, 2, 3])
@pytest.mark.parametrize("on_gpu", [False, True])
def test_out_of_core_equals_resident(orc, slots, on_gpu):
    base, q = _data(901), _data(902, 300)
    rng = _rng(orc)
    base_t = torch.from_numpy(base).cuda() if on_gpu else base
    ref = _build(base_t, rng, 0, on_gpu=on_gpu)
    ooc = _build(base_t, rng, slots, on_gpu=on_gpu)
    for a, b in zip(_graphs(ref, 4), _graphs(ooc, 4)):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    for nq in (300, 1):
        r = ref.query(q[:nq], 10, 0.7, 200)
        o = ooc.query(q[:nq], 10, 0.7, 200)
        assert torch.equal(r[0], o[0]) and torch.equal(r[1], o[1])
    # again: the slots now hold the LAST shards, the first ones come back from the host buffers
    o2 = ooc.query(q, 10, 0.7, 200)
    assert torch.equal(o2[0], ref.query(q, 10, 0.7, 200)[0])
    # exact brute force needs the whole base for the call
    rb, ob = ref.bf_query(q, 10), ooc.bf_query(q, 10)
    assert torch.equal(rb[0], ob[0]) and torch.equal(rb[1], ob[1])
    # batches in flight need resident shards
    with pytest.raises(RuntimeError, match="resident"):
        ooc.query_async(torch.from_numpy(q).cuda(), 10, 0.7, 200)
    # counters work shard by shard as well
    ooc.set_collect_counters(True)
    ref.set_collect_counters(True)
    ooc.query(q, 10, 0.7, 200)
    ref.set_prescreen(False)       # (out-of-core shards have no pre-screen copy: same counters)
    ref.query(q, 10, 0.7, 200)
    assert ooc.last_query_counters() == ref.last_query_counters()


@pytest.mark.parametrize("tier", ["host", "disk"])
def test_out_of_core_equals_oracle(orc, tmp_path, tier):
    """The swapping handle against the ORACLE, not only against a resident HIP handle (round-4
    verdict: a self-comparison proves nothing about the mode's results): every shard's graph pool
    equals `orc.build` of that slice of the base with the same selection numbers, and the query
    result equals `orc.query` per shard on the oracle's OWN graphs + the reference's per-GPU sort
    and ResultMerger (gpu_instance.cu:745-790, result_merger.cpp:51-149) -- for graphs waiting in
    page-locked host buffers and for graphs read back from part files."""
    base, q = _data(941), _data(942, 200)
    rng_full = orc.make_rng(N_SHARD, 5)
    kw = {}
    if tier == "disk":
        cfg0 = orc.graph_config(N_SHARD, D, K)
        pool_bytes = (cfg0.N_all * K + 2 * cfg0.ST_all) * 4 + 8
        kw = dict(workdir=tmp_path / "parts", cpu_limit=int(1.5 * pool_bytes))
    ooc = _build(base, rng_full[:3], 2, **kw)
    ids, d = ooc.query(q, 10, 0.7, 200)
    rows_i, rows_d = [], []
    for s, (g, tr, stats) in enumerate(_graphs(ooc, 4)):
        sl = base[s * N_SHARD:(s + 1) * N_SHARD]
        o_cfg, o_graph, o_tr, o_sel, o_stats = orc.build(sl, K, 0.5, 1, rng=rng_full)
        assert np.array_equal(g, o_graph[:g.shape[0]]), f"shard {s}: graph differs from the oracle's"
        assert np.array_equal(tr, o_tr[:tr.size]), f"shard {s}: translation differs"
        assert stats.tobytes() == o_stats.tobytes(), f"shard {s}: nn1 stats differ"
        start = o_tr[o_cfg.STs_offsets[3]:o_cfg.STs_offsets[3] + o_cfg.Ns[3]]
        o = orc.query(sl, q, o_graph[:N_SHARD], start, o_stats, 10, 0.7, 200)
        rows_i.append(o[0] + s * N_SHARD)
        rows_d.append(o[1])
    si, sd = orc.sort_shard_results(np.concatenate(rows_i, 1), np.concatenate(rows_d, 1))
    r_ids, r_d = orc.merge_results([si], [sd], 10, 4, N_SHARD)
    assert np.array_equal(d.numpy(), r_d)
    uniq = np.ones_like(r_d, bool)          # (order among exactly equal distances of different
    uniq[:, 1:] &= r_d[:, 1:] != r_d[:, :-1]   # shards is implementation-defined in the reference)
    uniq[:, :-1] &= r_d[:, :-1] != r_d[:, 1:]
    assert np.array_equal(ids.numpy()[uniq], r_ids[uniq])
    # and once more after the slots have been recycled
    ids2, d2 = ooc.query(q, 10, 0.7, 200)
    assert torch.equal(ids2, ids) and torch.equal(d2, d)
    # exact brute force of the swapping handle == the oracle's over the whole base
    b_ids, b_d = ooc.bf_query(q, 10)
    o_ids, o_d = orc.bf_query(base, q, 10)
    assert np.array_equal(b_ids.numpy(), o_ids) and np.array_equal(b_d.numpy(), o_d)


def test_out_of_core_disk_tier_and_store_load(orc, tmp_path):
    """one host buffer for four shards (ggnn_set_cpu_memory_limit): the graph parts live in
    part_<shard>.ggnn files of the working directory and are read back when a shard is needed;
    store() / load() interoperate with an all-resident handle"""
    import ggnn_amd as ggnn
    from ggnn_amd import _lib
    base, q = _data(911), _data(912, 200)
    rng = _rng(orc)
    ref = _build(base, rng, 0)
    g0 = ref.get_graph(0).config
    pool_bytes = (g0["N_all"] * K + 2 * g0["ST_all"]) * 4 + 8
    work = tmp_path / "ooc"
    ooc = _build(base, rng, 2, workdir=work, cpu_limit=int(1.5 * pool_bytes))
    files = sorted(os.listdir(work))
    assert files == [f"part_{s}.ggnn" for s in range(4)]
    assert all(os.path.getsize(work / f) == pool_bytes for f in files)
    r, o = ref.query(q, 10, 0.7, 200), ooc.query(q, 10, 0.7, 200)
    assert torch.equal(r[0], o[0]) and torch.equal(r[1], o[1])
    ooc.store()                                     # nothing left to write: everything is on disk
    # an all-resident handle loads what the out-of-core one wrote, and the other way round
    res2 = ggnn.GGNN()
    res2.set_base(torch.from_numpy(base)); res2.set_shard_size(N_SHARD)
    res2.set_working_directory(work); res2.load(K)
    assert torch.equal(res2.query(q, 10, 0.7, 200)[0], r[0])
    work2 = tmp_path / "resident"
    ref.set_working_directory(work2); ref.store()
    ooc2 = ggnn.GGNN()
    ooc2.set_base(torch.from_numpy(base)); ooc2.set_shard_size(N_SHARD)
    ooc2.set_working_directory(work2)
    with _lib.hooks(RESIDENT_SHARDS=1):
        ooc2.load(K)
    o2 = ooc2.query(q, 10, 0.7, 200)
    assert torch.equal(o2[0], r[0]) and torch.equal(o2[1], r[1])
    # a too small limit is refused like in the reference
    bad = ggnn.GGNN()
    bad.set_base(torch.from_numpy(base)); bad.set_shard_size(N_SHARD)
    bad.set_cpu_memory_limit(pool_bytes // 2)
    with _lib.hooks(RESIDENT_SHARDS=1), pytest.raises(MemoryError, match="single shard"):
        bad.build(K, 0.5, 1)


def test_out_of_core_two_device_contexts(orc):
    """several GPUs per handle, each with fewer slots than shards (both contexts on device 0)"""
    base, q = _data(921), _data(922, 150)
    rng = _rng(orc)
    ref = _build(base, rng, 0, gpus=(0, 0))
    ooc = _build(base, rng, 1, gpus=(0, 0))
    r, o = ref.query(q, 10, 0.7, 200), ooc.query(q, 10, 0.7, 200)
    assert torch.equal(r[0], o[0]) and torch.equal(r[1], o[1])
    assert ooc.last_query_parts() == 1


def test_all_shards_resident_by_default():
    """288 GB: a multi-shard handle never swaps unless it has to"""
    import ggnn_amd as ggnn
    base = _data(931)
    eng = ggnn.GGNN()
    eng.set_base(torch.from_numpy(base)); eng.set_shard_size(N_SHARD)
    eng.build(K, 0.5, 1)
    qd = torch.from_numpy(_data(932, 64)).cuda()
    eng.set_return_results_on_gpu(True)
    t = eng.query_async(qd, 10, 0.7, 100)     # only possible with every shard resident
    eng.synchronize()
    assert t.done
