"""ggnn_amd.benchmark: the flag-compatible counterpart of the reference's ggnn_benchmark driver
(examples/cpp-and-cuda/ggnn_benchmark.cpp:37-206)."""
import io
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")


def test_flags_and_defaults_match_the_reference():
    from ggnn_amd import benchmark
    args = benchmark.build_parser().parse_args([])
    # DEFINE_* lines 37-50 of ggnn_benchmark.cpp
    expected = dict(base="", subset=0, query="", gt="", graph_dir="", k_build=24, tau_build=0.5,
                    refinement_iterations=2, k_query=10, max_iterations=200, measure="euclidean",
                    shard_size=0, gpu_ids="0", grid_search=False)
    assert vars(args) == expected
    args = benchmark.build_parser().parse_args(
        ["--base=b.fvecs", "--query", "q.bvecs", "--gpu_ids", "0 1 2", "--grid_search",
         "--tau_build=0.6", "--shard_size", "125000000"])
    assert args.base == "b.fvecs" and args.query == "q.bvecs" and args.gpu_ids == "0 1 2"
    assert args.grid_search and args.tau_build == 0.6 and args.shard_size == 125000000


def test_tau_schedule():
    from ggnn_amd import benchmark
    assert benchmark.tau_schedule(False) == [0.34, 0.41, 0.51, 0.64]
    grid = benchmark.tau_schedule(True)  # ggnn_benchmark.cpp:189-192
    assert len(grid) == 84
    assert grid[0] == 0.0 and abs(grid[69] - 0.69) < 1e-12
    assert abs(grid[70] - 0.7) < 1e-12 and abs(grid[-1] - 2.0) < 1e-12


def test_measure_and_dataset_type_by_extension(tmp_path):
    from ggnn_amd import api, benchmark
    assert benchmark.parse_measure("euclidean") == api.DistanceMeasure.Euclidean
    assert benchmark.parse_measure("cosine") == api.DistanceMeasure.Cosine
    with pytest.raises(SystemExit):
        benchmark.parse_measure("manhattan")
    f = np.arange(12, dtype=np.float32).reshape(3, 4)
    api.FloatDataset(f).store(tmp_path / "x.fvecs")
    api.UCharDataset(f.astype(np.uint8)).store(tmp_path / "x.bvecs")
    a = benchmark.load_generic(str(tmp_path / "x.fvecs"))
    b = benchmark.load_generic(str(tmp_path / "x.bvecs"), num=2)
    assert isinstance(a, api.FloatDataset) and a.N == 3 and a.D == 4
    assert isinstance(b, api.UCharDataset) and b.N == 2
    with pytest.raises(RuntimeError):
        benchmark.load_generic(str(tmp_path / "x.txt"))
    with pytest.raises(SystemExit):
        benchmark.main(["--base", str(tmp_path / "missing.fvecs"), "--query",
                        str(tmp_path / "x.fvecs")], out=io.StringIO())


@pytest.mark.gpu
def test_benchmark_end_to_end(tmp_path):
    from ggnn_amd import api, benchmark
    # SIFT-like structure (16-dimensional latent), the reference's tau values are tuned for SIFT
    rng = np.random.default_rng(3)
    mix = rng.normal(size=(16, 64)) * 10.0
    base = np.clip(np.rint(128 + rng.normal(size=(20000, 16)) @ mix), 0, 255).astype(np.float32)
    query = np.clip(np.rint(128 + rng.normal(size=(500, 16)) @ mix), 0, 255).astype(np.float32)
    api.FloatDataset(base).store(tmp_path / "base.fvecs")
    api.FloatDataset(query).store(tmp_path / "query.fvecs")
    graph_dir = tmp_path / "graphs"
    graph_dir.mkdir()
    argv = ["--base", str(tmp_path / "base.fvecs"), "--query", str(tmp_path / "query.fvecs"),
            "--gt", str(tmp_path / "gt.ivecs"), "--graph_dir", str(graph_dir)]
    out = io.StringIO()
    first = benchmark.main(argv, out=out)
    assert os.path.isfile(graph_dir / "part_0.ggnn") and os.path.isfile(tmp_path / "gt.ivecs")
    assert "exporting brute-forced ground truth data." in out.getvalue()
    assert [t for t, _, _ in first] == [0.34, 0.41, 0.51, 0.64]
    assert first[-1][2].c1 > 0.9 and first[-1][2].r_k_query > 0.9
    gt = api.IntDataset.load(tmp_path / "gt.ivecs")
    assert gt.N == 500 and gt.D == 100  # bfQuery's default KGT
    # second run: graph and ground truth come from the files, results are the same
    out2 = io.StringIO()
    second = benchmark.main(argv, out=out2)
    assert "loaded the graph" in out2.getvalue() and "exporting" not in out2.getvalue()
    for (_, _, a), (_, _, b) in zip(first, second):
        assert a.c1 == b.c1 and a.r_k_query == b.r_k_query
