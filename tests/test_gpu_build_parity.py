"""the build / refine SCHEDULE (graph_construction.cu:128-147: merge / top per
(top, btm) pair, nn1 statistics on layer 0, select after top(l) for l < 3 with its own random
numbers :163-187, graph_buffer -> graph copy :292-295, sym + sym_buffer_merge after every merge
with the buffer memsets :298-379, refine = merge(3, l) + sym(l) for l = 2, 1, 0) compared
END TO END with the oracle's `orc_build`.

The reference's build is not reproducible (cuRAND stream; sym allocates inverse-link slots with
atomics and reads sym_buffer rows other blocks write).  `ggnn_set_build_hooks` pins the two
sources: the selection numbers are injected ([3][N], the same array the oracle reads) and sym
is launched one point at a time in ascending order -- the serialisation the oracle executes.
Everything else is the production path (merge with the pre-screen, nn1 statistics, copies).

Data: integer values that are multiples of 5 in [0, 255].  Squared L2 distances are exact in
float32 as for S-int, and so is sym's half point `q + (0.5 - 0.1)(start - q)`
(simple_knn_sym_cache.cuh:159-177): 0.4f * 5m rounds to 2m exactly, so the half-point distance
-- which is inexact on general integer data and decides which links sym requests -- is an exact
integer on both sides; no decision depends on a summation order.  Bar: graph (all four layers),
translation, selection and nn1_stats equal BIT FOR BIT.
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


def mult5_data(N, D, seed, dtype):
    v = np.random.default_rng(seed).integers(0, 52, (N, D)) * 5
    return v.astype(dtype)


def engine_build(base, K, tau, refine, rng, measure=None, prescreen=True):
    import ggnn_amd as ggnn
    eng = ggnn.GGNN()
    eng.set_base(torch.from_numpy(base))
    eng.set_prescreen(prescreen)
    eng.set_build_hooks(rng[:3], serial_sym=True)
    if measure is None:
        eng.build(K, tau, refine)
    else:
        eng.build(K, tau, refine, measure)
    g = eng.get_graph(0)
    cfg = g.config
    graph = np.concatenate([g.graph[l].view.numpy().reshape(-1, K) for l in range(4)])
    tr = np.concatenate([g.translation[l].view.numpy().reshape(-1) for l in range(1, 4)])
    sel = np.concatenate([g.selection[l].view.numpy().reshape(-1) for l in range(1, 4)])
    stats = g.nn1_stats.view.numpy().reshape(-1).copy()
    return cfg, graph, tr, sel, stats, eng


def compare(orc, base, K, tau, refine, seed, prescreen=True):
    N = base.shape[0]
    rng = orc.make_rng(N, seed)
    o_cfg, o_graph, o_tr, o_sel, o_stats = orc.build(base, K, tau, refine, rng=rng)
    cfg, graph, tr, sel, stats, eng = engine_build(base, K, tau, refine, rng, prescreen=prescreen)
    assert cfg["Ns"] == list(o_cfg.Ns) and cfg["G"] == o_cfg.G and cfg["SG"] == o_cfg.SG
    assert stats.tobytes() == o_stats.tobytes(), (stats, o_stats)
    assert np.array_equal(tr, o_tr[:tr.size]), "translation differs"
    assert np.array_equal(sel, o_sel[:sel.size]), "selection differs"
    for l in range(4):
        a, b = o_cfg.Ns_offsets[l], o_cfg.Ns_offsets[l] + o_cfg.Ns[l]
        bad = np.nonzero((graph[a:b] != o_graph[a:b]).any(1))[0]
        assert bad.size == 0, f"layer {l}: {bad.size} of {b - a} rows differ, first {bad[:5]}"
    return eng


@pytest.mark.parametrize("refine", [0, 1, 2])
def test_build_schedule_bit_exact_f32(orc, refine):
    base = mult5_data(6000, 128, 1234 + refine, np.float32)
    compare(orc, base, 24, 0.5, refine, seed=7 + refine)


def test_build_schedule_bit_exact_without_prescreen(orc):
    """same build through the plain float kernels (the pre-screen changes no result)"""
    base = mult5_data(5000, 128, 77, np.float32)
    compare(orc, base, 24, 0.5, 1, seed=3, prescreen=False)


def test_build_schedule_bit_exact_d96(orc):
    """DEEP-shaped rows (384 B, three 16-byte chunks per lane group) and an odd N"""
    base = mult5_data(7777, 96, 4321, np.float32)
    compare(orc, base, 24, 0.5, 2, seed=11)


def test_build_schedule_bit_exact_u8(orc):
    base = mult5_data(6000, 128, 99, np.uint8)
    compare(orc, base, 24, 0.5, 2, seed=5)


def test_build_schedule_bit_exact_other_k_tau(orc):
    """KBuild 20 (KF 10, S 32) with a wider slack"""
    base = mult5_data(5000, 64, 31, np.float32)
    compare(orc, base, 20, 0.7, 1, seed=13)


def test_build_schedule_larger_graph_and_query(orc):
    """20k points (G = 9, SG = 3, SG_off = 5: uneven promotion per segment), then the oracle's
    traversal over the engine-built graph equals the engine's query"""
    base = mult5_data(20000, 128, 2024, np.float32)
    eng = compare(orc, base, 24, 0.5, 2, seed=17)
    q = mult5_data(300, 128, 555, np.float32)
    ids, d = eng.query(torch.from_numpy(q), 10, 0.64, 400)
    g = eng.get_graph(0)
    o_ids, o_d = orc.query(base, q, g.graph[0].view.numpy(),
                           g.translation[3].view.numpy().reshape(-1),
                           g.nn1_stats.view.numpy().reshape(-1), 10, 0.64, 400)
    assert np.array_equal(ids.numpy(), o_ids) and np.array_equal(d.numpy(), o_d)


def test_build_hooks_reset_and_default_build_differs_only_by_schedule(orc):
    """hooks off again -> the production build (own generator, parallel sym) still gives a
    valid graph of the same shape; the hooked build is repeatable"""
    import ggnn_amd as ggnn
    base = mult5_data(4096, 128, 8, np.float32)
    rng = orc.make_rng(4096, 21)
    a = engine_build(base, 24, 0.5, 1, rng)
    b = engine_build(base, 24, 0.5, 1, rng)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    eng = ggnn.GGNN()
    eng.set_base(torch.from_numpy(base))
    eng.set_build_hooks(rng[:3], serial_sym=True)
    eng.set_build_hooks(None, serial_sym=False)
    eng.build(24, 0.5, 1)
    g = eng.get_graph(0).graph[0].view.numpy()
    assert g.min() >= 0 and g.max() < 4096
