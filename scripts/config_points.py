"""Per-shard operating points of the secondary BASELINE configurations on one MI355X (the numbers
quoted in DESIGN.md section 4): GIST1M shape (1M x 960 f32, cosine), DEEP100M/8 shard (12.5M x 96
f32), SIFT1M shape as uint8, SIFT1B/8 shard (125M x 128 uint8).  Writes one JSON document.
    python scripts/config_points.py out.json [gist deep u8 sift1b]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ggnn_amd as ggnn
from bench import synthetic, recall_at_k
ggnn.set_log_level(-1)
dev = torch.device("cuda", 0)
out_path = sys.argv[1]
which = sys.argv[2:] or ["gist", "deep", "u8", "sift1b"]
doc = {}


def run(name, N, D, measure, kind, points, dtype=torch.float32, nq=10_000):
    base = torch.empty((N, D), dtype=dtype, device=dev)
    for lo in range(0, N, 5_000_000):
        hi = min(N, lo + 5_000_000)
        base[lo:hi] = synthetic(kind, hi - lo, D, 1234 + lo, dev).to(dtype)
    query = synthetic(kind, nq, D, 4321, dev).to(dtype)
    eng = ggnn.GGNN(); eng.set_base_reference(base); eng.set_return_results_on_gpu(True)
    eng.build(24, 0.5, 2, measure)
    r = {"N": N, "D": D, "dtype": str(dtype).split(".")[-1], "measure": "cosine" if measure else "L2",
         "dataset": kind, "n_query": nq, "graph_build_s": eng.last_timing_ms()["build_ms"] / 1e3}
    gt, _ = eng.bf_query(query, 10, measure)
    r["bf_query_ms"] = eng.last_timing_ms()["bf_query_ms"]
    r["bf_rescanned"] = eng.last_bf_query_rescanned()
    r["points"] = []
    for tau, it in points:
        for _ in range(3):
            ids, d = eng.query(query, 10, tau, it, measure)
        ms = eng.last_timing_ms()["query_ms"]
        p = {"tau_query": tau, "max_iterations": it, "query_kernel_ms": ms,
             "queries_per_s": nq / (ms * 1e-3), "recall_at_10": recall_at_k(ids, gt)}
        if dtype == torch.float32:
            eng.set_prescreen(False)
            for _ in range(2):
                ids2, d2 = eng.query(query, 10, tau, it, measure)
            p["without_prescreen_ms"] = eng.last_timing_ms()["query_ms"]
            p["prescreen_bit_identical"] = bool(torch.equal(ids, ids2) and torch.equal(d, d2))
            eng.set_prescreen(True)
        r["points"].append(p)
    doc[name] = r
    print(name, json.dumps(r), flush=True)
    json.dump(doc, open(out_path, "w"), indent=1)
    del eng, base
    torch.cuda.empty_cache()


if "gist" in which:
    run("GIST1M-shaped (C3)", 1_000_000, 960, 1, "lowrank16", [(0.9, 200), (1.0, 400)])
if "deep" in which:
    run("DEEP100M/8 shard (C4)", 12_500_000, 96, 0, "lowrank16", [(0.9, 200), (1.0, 400), (1.2, 400)])
if "u8" in which:
    run("SIFT1M-shaped uint8", 1_000_000, 128, 0, "lowrank16", [(0.9, 175), (1.0, 400)], torch.uint8)
if "sift1b" in which:
    run("SIFT1B/8 shard (C5)", 125_000_000, 128, 0, "lowrank16",
        [(1.0, 400), (1.5, 400)], torch.uint8)
