"""CPU tests of the C-ABI boundary and the host-side mirror of the reference interface: the
library loads, exports every symbol include/ggnn_c.h declares, host-only entry points agree
with the oracle, and API misuse maps to the reference's exceptions.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from ggnn_amd import _lib
    return _lib.lib()


def declared_functions():
    src = open(os.path.join(ROOT, "include", "ggnn_c.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ggnn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib):
    from ggnn_amd import _lib
    names = declared_functions()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in ggnn_c.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(names)
    assert b"gfx950" in lib.ggnn_version()


def test_graph_config_matches_oracle(lib, orc):
    from ggnn_amd._lib import GraphConfig
    for N, D, K in [(10_000, 128, 24), (1_000_000, 128, 24), (12_500_000, 96, 24),
                    (125_000_000, 128, 24), (25_000, 128, 24), (4096, 64, 8), (77_777, 32, 40)]:
        c = GraphConfig()
        assert lib.ggnn_graph_config_init(N, D, K, C.byref(c)) == 0
        assert c.as_dict() == orc.graph_config(N, D, K).as_dict()
    c = GraphConfig()
    assert lib.ggnn_graph_config_init(1000, 8, 1, C.byref(c)) == 1   # KBuild < 2
    assert lib.ggnn_graph_config_init(10, 8, 24, C.byref(c)) == 1     # base too small


def test_query_sizing_matches_oracle(lib, orc):
    for D, K, it in [(128, 10, 200), (128, 10, 400), (960, 10, 400), (128, 100, 2000),
                     (96, 1, 64), (128, 47, 400), (128, 48, 400)]:
        cache, sorted_ = C.c_uint32(), C.c_uint32()
        assert lib.ggnn_query_sizing(D, K, it, C.byref(cache), C.byref(sorted_)) == 0
        s = orc.query_sizing(D, K, it)
        assert (cache.value, sorted_.value) == (s.cache_size, s.sorted_size)
    cache, sorted_ = C.c_uint32(), C.c_uint32()
    assert lib.ggnn_query_sizing(128, 6001, 400, C.byref(cache), C.byref(sorted_)) == 1
    assert lib.ggnn_query_sizing(128, 10, 8193, C.byref(cache), C.byref(sorted_)) == 1


def test_error_conventions():
    """ggnn.cu:96,150,209,282,336: misuse raises RuntimeError / IndexError like the reference."""
    import ggnn_amd as ggnn
    g = ggnn.GGNN()
    with pytest.raises(RuntimeError, match="base needs to be set"):
        g.build(24, 0.5)
    with pytest.raises(RuntimeError, match="no graph to query"):
        g.query(np.zeros((2, 8), np.float32), 10, 0.5)
    with pytest.raises(RuntimeError, match="No graph has been built"):
        g.get_graph()
    with pytest.raises(RuntimeError, match="no graph to store"):
        g.store()
    with pytest.raises(IndexError, match="Invalid GPU index"):
        g.set_gpus([-1])
    with pytest.raises(IndexError):
        g.set_gpus([10_000])
    with pytest.raises(TypeError):
        g.set_base(np.zeros((4, 4), np.float64))
    with pytest.raises(TypeError):
        g.set_base(np.zeros(16, np.float32))
    g.set_base(np.zeros((64, 8), np.float32))
    with pytest.raises(RuntimeError, match="different data type"):
        g.set_base(np.zeros((64, 8), np.uint8))


def test_module_surface():
    """names / defaults of the nanobind module (nanobind.cu:151-300)."""
    import inspect

    import ggnn
    import ggnn_amd
    for name in ("GGNN", "DistanceMeasure", "FloatDataset", "UCharDataset", "IntDataset",
                 "Evaluator", "Evaluation", "Graph", "set_log_level"):
        assert hasattr(ggnn, name) and getattr(ggnn, name) is getattr(ggnn_amd, name)
    assert int(ggnn.DistanceMeasure.Euclidean) == 0 and int(ggnn.DistanceMeasure.Cosine) == 1
    sig = inspect.signature(ggnn.GGNN.build)
    assert list(sig.parameters)[1:] == ["k_build", "tau_build", "refinement_iterations", "measure"]
    assert sig.parameters["refinement_iterations"].default == 2
    sig = inspect.signature(ggnn.GGNN.query)
    assert list(sig.parameters)[1:] == ["query", "k_query", "tau_query", "max_iterations",
                                        "measure"]
    assert sig.parameters["max_iterations"].default == 400
    sig = inspect.signature(ggnn.GGNN.bf_query)
    assert list(sig.parameters)[1:] == ["query", "k_gt", "measure"]
    assert sig.parameters["k_gt"].default == 100
    assert list(inspect.signature(ggnn.GGNN.get_graph).parameters)[1:] == ["on_gpu_shard_id"]
    assert list(inspect.signature(ggnn.GGNN.set_shard_size).parameters)[1:] == ["n_shard"]
    assert inspect.signature(
        ggnn.GGNN.set_return_results_on_gpu).parameters["return_results_on_gpu"].default is True
    assert list(inspect.signature(ggnn.Evaluator.__init__).parameters)[1:] == [
        "base", "query", "gt", "k_query", "measure"]


def test_dataset_xvecs_roundtrip(tmp_path):
    import ggnn_amd as ggnn
    a = np.random.default_rng(0).random((37, 24), dtype=np.float32)
    ds = ggnn.FloatDataset(a)
    assert (ds.N, ds.D, ds.numel(), ds.device) == (37, 24, 37 * 24, "cpu")
    f = tmp_path / "x.fvecs"
    ds.store(str(f))
    assert os.path.getsize(f) == 37 * (4 + 24 * 4)      # dataset.cu: uint32 D + D values / row
    back = ggnn.FloatDataset.load(str(f))
    assert np.array_equal(back.view.numpy(), a)
    part = ggnn.FloatDataset.load(str(f), 5, 10)
    assert np.array_equal(part.view.numpy(), a[5:15])
    b = np.random.default_rng(1).integers(0, 256, (11, 16)).astype(np.uint8)
    fb = tmp_path / "x.bvecs"
    ggnn.UCharDataset(b).store(str(fb))
    assert np.array_equal(ggnn.UCharDataset.load(str(fb)).view.numpy(), b)


@pytest.mark.parametrize("measure", [0, 1])
def test_evaluator_matches_oracle(orc, measure):
    """ggnn.Evaluator (host logic) against the oracle's restatement of eval.cpp, with duplicated
    base rows so that the duplicate-aware counters differ from the plain ones."""
    import ggnn_amd as ggnn
    r = np.random.default_rng(3)
    base = r.integers(0, 4, (300, 8)).astype(np.float32) + 1
    base[100:200] = base[:100]                      # duplicates
    query = r.integers(0, 4, (40, 8)).astype(np.float32) + 1
    gt, _ = orc.bf_query(base, query, 20, measure)
    res = gt[:, :10].copy()
    res[::3, 0] = gt[::3, 1]                        # perturb some results
    res[1::4, 5] = 299
    for K in (10, 5):
        ev = ggnn.Evaluator(base, query, gt, K, ggnn.DistanceMeasure(measure))
        got = ev.evaluate_results(res[:, :K].copy())
        ref = orc.evaluate(base, query, gt, K, res[:, :K].copy(), measure)
        for name in ref:
            np.testing.assert_allclose(getattr(got, name), ref[name], rtol=1e-6, err_msg=name)
        assert "c@1 (=r@1):" in repr(got) and f"r@{K}:" in repr(got)
    ev = ggnn.Evaluator(np.zeros((0, 8), np.float32), np.zeros((0, 8), np.float32), gt, 10)
    got = ev.evaluate_results(res)
    ref = orc.evaluate(None, None, gt, 10, res)
    assert np.isnan(got.c1_dup) and "(duplicates unknown)" in repr(got)
    np.testing.assert_allclose(got.c_k_query, ref["c_k_query"], rtol=1e-6)


def test_reference_python_examples_api_conformance():
    """Every module attribute, GGNN/Evaluator method and keyword argument used by the reference's
    examples/python/*.py exists with the same name here (checked by parsing the examples; only
    where the reference tree is mounted)."""
    import ast
    import glob
    import inspect

    import ggnn
    files = sorted(glob.glob("/root/reference/examples/python/*.py"))
    if not files:
        pytest.skip("reference tree not mounted")
    checked = 0
    for path in files:
        tree = ast.parse(open(path).read())
        instances = {}   # variable name -> class
        for node in ast.walk(tree):
            if isinstance(node, ast.Assign) and isinstance(node.value, ast.Call):
                f = node.value.func
                if isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name) and f.value.id == "ggnn":
                    for t in node.targets:
                        if isinstance(t, ast.Name):
                            instances[t.id] = getattr(ggnn, f.attr, None)
        for node in ast.walk(tree):
            if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name):
                if node.value.id == "ggnn":
                    assert hasattr(ggnn, node.attr), f"{path}: ggnn.{node.attr} missing"
                    checked += 1
            if not isinstance(node, ast.Call):
                continue
            f = node.func
            target = None
            if isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name):
                if f.value.id == "ggnn":
                    target = getattr(ggnn, f.attr)
                    target = target.__init__ if inspect.isclass(target) and f.attr != "DistanceMeasure" else target
                elif f.value.id in instances and inspect.isclass(instances[f.value.id]):
                    assert hasattr(instances[f.value.id], f.attr), f"{path}: .{f.attr} missing"
                    target = getattr(instances[f.value.id], f.attr)
            elif (isinstance(f, ast.Attribute) and isinstance(f.value, ast.Attribute) and
                  isinstance(f.value.value, ast.Name) and f.value.value.id == "ggnn"):
                cls = getattr(ggnn, f.value.attr)          # e.g. ggnn.FloatDataset.load
                assert hasattr(cls, f.attr), f"{path}: ggnn.{f.value.attr}.{f.attr} missing"
                target = getattr(cls, f.attr)
            if target is None or not node.keywords:
                continue
            try:
                params = inspect.signature(target).parameters
            except (TypeError, ValueError):
                continue
            if any(p.kind == p.VAR_KEYWORD for p in params.values()):
                continue
            for kw in node.keywords:
                assert kw.arg in params, f"{path}: keyword {kw.arg} not accepted by {f.attr}"
                checked += 1
    assert checked > 30


def test_prescreen_code_rows_never_straddle_more_lines_than_needed(lib):
    """ggnn_prescreen_sizes (host only): the code-row pitch is a power of two up to 64 bytes and a
    multiple of 64 above, always >= D and < 2 D -- a 96-byte pitch cost 1.40x the algorithmic
    bytes on the fabric (profiles/r03_d96_*)."""
    import ctypes as C
    dc, pf, sf = C.c_uint32(), C.c_size_t(), C.c_size_t()
    for D in range(4, 4097, 4):
        assert lib.ggnn_prescreen_sizes(1000, D, 0, C.byref(dc), C.byref(pf), C.byref(sf)) == 0
        v = dc.value
        assert D <= v < max(2 * D, 17) and v % 16 == 0
        if D <= 64:
            assert v & (v - 1) == 0            # several whole rows per 128-byte line
        else:
            assert v % 64 == 0                 # rows start on 64-byte granules
        assert pf.value == 8 + v
    # D not a multiple of 4 is refused (float4 accesses)
    assert lib.ggnn_prescreen_sizes(1000, 65, 0, C.byref(dc), C.byref(pf), C.byref(sf)) != 0


def test_hooks_api(lib, monkeypatch):
    """ggnn_set_hook / ggnn_get_hook / ggnn_reset_hook; the environment only counts while
    GGNN_TEST_HOOKS=1 (conftest sets it for the test session)."""
    from ggnn_amd import _lib
    header = open(os.path.join(ROOT, "include", "ggnn_c.h")).read()
    names = ["PRESCREEN", "EXCHANGE", "SYM_PRESCREEN", "SHARD_OVERLAP", "VIS_SLOTS", "VIS_TAG_SET", "QUERY_SPLIT", "RESIDENT_SHARDS", "XCD_MAP",
             "BF_POOL_KEEP_MB", "BF_NO_I8", "BF_I8_V1", "BF_SLICES", "BF_NO_CENTER", "BF_TILES",
             "BF_I8_NOSHARE", "BF_I8_RANKS", "BF_SCAN", "RCCL_FAIL_AFTER", "QUERY_EARLY", "MERGE_EARLY", "QUERY_LDS_PAD", "QUERY_GLOBAL_RING", "BF_I8_REFRESH", "BF_I8_SEED", "MERGE_COUNTING"]
    for n in names:
        assert re.search(r"\*\s+" + n + r"\s", header), f"hook {n} is not documented in ggnn_c.h"
        _lib.get_hook(n)
    # every getenv of the product library goes through the registry
    src_dir = os.path.join(ROOT, "ggnn_amd", "csrc")
    for f in os.listdir(src_dir):
        if f.endswith((".hip", ".cpp", ".hpp")) and f != "hooks.cpp":
            text = open(os.path.join(src_dir, f)).read()
            assert "getenv" not in text, f"{f} reads the environment directly"
    assert _lib.get_hook("VIS_SLOTS") == 8 and _lib.get_hook("EXCHANGE") == 0
    monkeypatch.setenv("GGNN_VIS_SLOTS", "2")
    monkeypatch.setenv("GGNN_EXCHANGE", "rccl")
    assert _lib.get_hook("VIS_SLOTS") == 2 and _lib.get_hook("GGNN_EXCHANGE") == 1
    monkeypatch.setenv("GGNN_TEST_HOOKS", "0")
    assert _lib.get_hook("VIS_SLOTS") == 8 and _lib.get_hook("EXCHANGE") == 0   # env ignored
    with _lib.hooks(VIS_SLOTS=4):
        assert _lib.get_hook("VIS_SLOTS") == 4                                   # setter always counts
    assert _lib.get_hook("VIS_SLOTS") == 8
    monkeypatch.setenv("GGNN_TEST_HOOKS", "1")
    with _lib.hooks(VIS_SLOTS=4):
        assert _lib.get_hook("VIS_SLOTS") == 4                                   # setter beats env
    assert _lib.get_hook("VIS_SLOTS") == 2
    with pytest.raises(ValueError):
        _lib.set_hook("NO_SUCH_HOOK", 1)
    v = C.c_int64()
    assert lib.ggnn_get_hook(b"NO_SUCH_HOOK", C.byref(v)) == 1
