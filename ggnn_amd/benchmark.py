"""Flag-compatible counterpart of the reference's benchmark driver.

    python -m ggnn_amd.benchmark --base sift_base.fvecs --query sift_query.fvecs \
        --gt sift_groundtruth.ivecs --graph_dir /tmp/graphs --k_build 24 --tau_build 0.5 \
        --refinement_iterations 2 --k_query 10 --max_iterations 200 --measure euclidean \
        --shard_size 0 --gpu_ids "0" [--grid_search]

Mirrors examples/cpp-and-cuda/ggnn_benchmark.cpp:37-206 of the reference: same flags (gflags
spelling, `--flag value` or `--flag=value`), same flow -- load base/query (type from the file
extension), load the graph when <graph_dir>/part_0.ggnn exists, otherwise build and store it;
load the ground truth when the file exists, otherwise brute-force it (and export it when --gt
names a file); query with tau_query 0.34 / 0.41 / 0.51 / 0.64, or the reference's grid of 84
values with --grid_search; print the Evaluator's report for every run.

Kept as in the reference: the brute-force ground truth is computed with bfQuery's default
arguments (100 neighbours, Euclidean) whatever --measure says (ggnn_benchmark.cpp:168).
"""
import argparse
import os
import sys
import time

from . import api


def build_parser():
    p = argparse.ArgumentParser(prog="ggnn_benchmark", description=__doc__.split("\n\n")[0])
    p.add_argument("--base", default="", help="Path to file with base vectors (fvecs/bvecs).")
    p.add_argument("--subset", type=int, default=0, help="Number of base vectors to use.")
    p.add_argument("--query", default="", help="Path to file with query vectors (fvecs/bvecs).")
    p.add_argument("--gt", default="", help="Path to file with groundtruth vectors (ivecs).")
    p.add_argument("--graph_dir", default="",
                   help="Directory to store and load ggnn graph files.")
    p.add_argument("--k_build", type=int, default=24,
                   help="Number of neighbors for graph construction")
    p.add_argument("--tau_build", type=float, default=0.5,
                   help="Search graph construction slack factor.")
    p.add_argument("--refinement_iterations", type=int, default=2,
                   help="Number of refinement iterations.")
    p.add_argument("--k_query", type=int, default=10, help="Number of neighbors to query for.")
    p.add_argument("--max_iterations", type=int, default=200,
                   help="Maximum number of search iterations per query.")
    p.add_argument("--measure", default="euclidean",
                   help="Distance measure. (euclidean or cosine)")
    p.add_argument("--shard_size", type=int, default=0, help="Number of vectors per shard.")
    p.add_argument("--gpu_ids", default="0", help="GPU ids, separated by spaces.")
    p.add_argument("--grid_search", action="store_true",
                   help="Perform queries for a wide range of parameters.")
    return p


def load_generic(path, num=None):
    """GenericDataset::load (dataset.cu:302-330): element type from the file extension."""
    ext = os.path.splitext(path)[1].lower()
    kw = {} if num is None else {"num": num}
    if ext == ".fvecs":
        return api.FloatDataset.load(path, **kw)
    if ext == ".bvecs":
        return api.UCharDataset.load(path, **kw)
    raise RuntimeError(f"unsupported file type (need .fvecs or .bvecs): {path}")


def parse_measure(name):
    if name == "euclidean":
        return api.DistanceMeasure.Euclidean
    if name == "cosine":
        return api.DistanceMeasure.Cosine
    raise SystemExit(f"invalid measure: {name}")


def tau_schedule(grid_search):
    """ggnn_benchmark.cpp:186-201"""
    if grid_search:
        return [i * 0.01 for i in range(70)] + [i * 0.1 for i in range(7, 21)]
    return [0.34, 0.41, 0.51, 0.64]


def main(argv=None, out=sys.stdout):
    args = build_parser().parse_args(argv)

    def log(msg):
        print(msg, file=out, flush=True)

    if not os.path.exists(args.base):
        raise SystemExit(f"File for base vectors has to exist: {args.base}")
    if not os.path.exists(args.query):
        raise SystemExit(f"File for query vectors has to exist: {args.query}")
    if args.tau_build < 0:
        raise SystemExit("tau_build has to be bigger or equal 0.")
    if args.refinement_iterations < 0:
        raise SystemExit("The number of refinement iterations has to be non-negative.")
    measure = parse_measure(args.measure)
    gpus = [int(tok) for tok in args.gpu_ids.split()]

    base = load_generic(args.base, args.subset if args.subset else None)
    query = load_generic(args.query)
    log(f"base: {base.N} x {base.D} ({args.base}), query: {query.N} x {query.D}")

    g = api.GGNN()
    g.set_working_directory(args.graph_dir)
    g.set_base_reference(base)
    g.set_gpus(gpus)
    g.set_shard_size(args.shard_size)

    part0 = os.path.join(args.graph_dir, "part_0.ggnn") if args.graph_dir else ""
    if part0 and os.path.isfile(part0):
        g.load(args.k_build)
        log(f"loaded the graph from {args.graph_dir}")
    else:
        t0 = time.perf_counter()
        g.build(args.k_build, args.tau_build, args.refinement_iterations, measure)
        log(f"build: {time.perf_counter() - t0:.3f} s wall, "
            f"{g.last_timing_ms()['build_ms'] / 1000.0:.3f} s on the GPU")
        if args.graph_dir:
            g.store()

    if args.gt and os.path.isfile(args.gt):
        gt = api.IntDataset.load(args.gt)
    else:
        ids, _ = g.bf_query(query)  # defaults, as the reference does
        gt = api.IntDataset._wrap(ids.cpu() if hasattr(ids, "cpu") else ids)
        if args.gt:
            log("exporting brute-forced ground truth data.")
            gt.store(args.gt)

    evaluator = api.Evaluator(base, query, gt, args.k_query, measure)
    reports = []
    log("--")
    log("grid-search:" if args.grid_search else
        "Querying for 90, 95, 99% R@1 (if running on SIFT1M with default parameters):")
    for tau in tau_schedule(args.grid_search):
        log("--")
        log(f"Query with tau_query {tau:g} max iterations {args.max_iterations}")
        ids, _ = g.query(query, args.k_query, tau, args.max_iterations, measure)
        ms = g.last_timing_ms()["query_ms"]
        report = evaluator.evaluate_results(ids)
        reports.append((tau, ms, report))
        log(f"{query.N} queries in {ms:.3f} ms ({query.N / ms * 1e3:,.0f} queries/s)")
        log(str(report))
    return reports


if __name__ == "__main__":
    main()
