"""bench.py pieces that need no GPU: stale-profile detection, speed-up bookkeeping, CLI defaults."""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _args(**kw):
    d = dict(dataset="lowrank16", n_base=1_000_000, dim=128, dtype="f32", measure="l2",
             n_query=10_000, k=10, k_build=24, tau_build=0.5, refine=2, tau_query=0.9,
             max_iters=175)
    d.update(kw)
    return types.SimpleNamespace(**d)


def test_workload_string_names_shape_dtype_measure():
    import bench
    w = bench.workload_string(_args())
    assert "SIFT1M-shaped 1000000x128 f32" in w and "cosine" not in w
    w = bench.workload_string(_args(dim=960, measure="cosine"))
    assert "1000000x960 f32 per GPU, cosine" in w and "SIFT1M-shaped" not in w
    assert "u8" in bench.workload_string(_args(dtype="u8"))


def test_committed_counter_passes_are_ignored_when_sources_changed(tmp_path, monkeypatch):
    import bench
    args = _args()
    prof = tmp_path / "profiles"
    prof.mkdir()
    kern = {"void ggnn_amd::query_kernel<float, 16, 2, 1, 0, ggnn_amd::Prescreen<8, 1, 0>, 1>":
            {"SQ_ACTIVE_INST_VALU": 6.0e8, "SQ_INSTS_VALU": 6.4e8,
             "FETCH_SIZE": {"avg_kb": 1000.0}, "WRITE_SIZE": {"avg_kb": 10.0}}}
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_sha", lambda sources=None: "aaaa")
    for suffix in ("_pmc_sq.json", "_pmc_hbm.json"):
        json.dump({"workload": bench.workload_string(args), "kernel_source_sha": "aaaa",
                   "kernels": kern}, open(prof / ("r99" + suffix), "w"))
    c, f, note = bench.pmc_sq(args, True)
    assert note is None and c["SQ_ACTIVE_INST_VALU"] == 6.0e8 and f.endswith("r99_pmc_sq.json")
    assert bench.pmc_traffic(args, True) == 2 * 1000.0 * 1024 + 10.0 * 1024
    # the sources moved on: both are refused, with a reason
    monkeypatch.setattr(bench, "kernel_source_sha", lambda sources=None: "bbbb")
    c, f, note = bench.pmc_sq(args, True)
    assert c is None and "changed" in note
    assert bench.pmc_traffic(args, True) is None
    # another workload never matches
    c, f, note = bench.pmc_sq(_args(dtype="u8"), False)
    assert c is None and "no committed" in note


def test_speedups_compare_like_with_like():
    import bench
    one = {"queries_per_s": 1.0e6, "saturated_batch": {"queries_per_s": 1.1e6},
           "pipelined_batches": {"queries_per_s": 1.2e6}}
    case = {"queries_per_s": 5.0e6, "saturated_batch": {"queries_per_s": 8.8e6}}
    s = bench.speedups(case, one)
    assert s == {"blocking": 5.0, "saturated_batch": 8.0}
    assert bench.speedups(case, {"error": "x"}) is None and bench.speedups(case, None) is None


def test_kernel_source_sha_is_stable_and_tracks_sources():
    import bench
    a = bench.kernel_source_sha()
    assert a == bench.kernel_source_sha() and len(a) == 16


def test_operating_points_resolve_per_shape_and_command_line_wins():
    """bench.py quotes `value` at a tuned (tau, iterations) per shape; explicit flags override"""
    import bench
    a = _args(tau_query=None, max_iters=None)
    bench.resolve_operating_point(a, 1)
    assert (a.tau_query, a.max_iters) == bench.OPERATING_POINTS[(1, 1_000_000, 128, "f32", "l2")][:2]
    assert a.operating_point_source.startswith("OPERATING_POINTS")
    # the N > 1 series: 8 shards of 12.5M x 96, tuned on MERGED recall
    b = _args(tau_query=None, max_iters=None, n_base=12_500_000, dim=96)
    bench.resolve_operating_point(b, bench.TOTAL_SHARDS)
    assert (b.tau_query, b.max_iters) == bench.OPERATING_POINTS[(8, 12_500_000, 96, "f32", "l2")][:2]
    one = _args(tau_query=None, max_iters=None, n_base=12_500_000, dim=96)
    bench.resolve_operating_point(one, 1)
    assert (one.tau_query, one.max_iters) != (b.tau_query, b.max_iters)   # a lone shard needs more
    # an untuned shape falls back, an explicit flag is kept
    c = _args(tau_query=None, max_iters=None, dim=960, measure="cosine")
    bench.resolve_operating_point(c, 1)
    assert (c.tau_query, c.max_iters) == bench.FALLBACK_POINT and "fallback" in c.operating_point_source
    d = _args(tau_query=1.0, max_iters=None)
    bench.resolve_operating_point(d, 1)
    assert d.tau_query == 1.0 and d.operating_point_source == "command line"
    # every tuned point states the recall it was confirmed at
    assert all(len(v) == 3 and v[2] for v in bench.OPERATING_POINTS.values())
