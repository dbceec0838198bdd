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
