"""Scale / shape exploration: DEEP100M shard (12.5M x 96), GIST1M shape (1M x 960), u8 SIFT."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ggnn_amd as ggnn
from bench import synthetic, recall_at_k
ggnn.set_log_level(-1)
dev = torch.device("cuda", 0)
def run(name, N, D, dtype, measure, tau, it, kind="lowrank16"):
    query = synthetic(kind, 10_000, D, 4321, dev)
    if dtype == "u8":
        # generate in chunks to keep the float staging small
        base = torch.empty((N, D), dtype=torch.uint8, device=dev)
        for lo in range(0, N, 5_000_000):
            hi = min(N, lo + 5_000_000)
            base[lo:hi] = synthetic(kind, hi - lo, D, 1234 + lo, dev).to(torch.uint8)
        query = query.to(torch.uint8)
    else:
        base = synthetic(kind, N, D, 1234, dev)
    eng = ggnn.GGNN(); eng.set_base_reference(base); eng.set_return_results_on_gpu(True)
    t = time.time(); eng.build(24, 0.5, 2, measure); tb = time.time() - t
    gt, _ = eng.bf_query(query, 10, measure); bf = eng.last_timing_ms()["bf_query_ms"]
    eng.set_collect_counters(True)
    for _ in range(3): ids, d = eng.query(query, 10, tau, it, measure)
    ms = eng.last_timing_ms()["query_ms"]; c = eng.last_query_counters()
    rowb = D * (1 if dtype == "u8" else 4)
    print(f"{name}: N={N} D={D} {dtype} build {tb:.2f}s  bf {bf:.0f} ms  query {ms:.2f} ms = {1e4/ms*1e3:,.0f} qps  recall@10={recall_at_k(ids, gt):.4f}  n_dist/q={c['n_dist']/1e4:.0f}  alg {c['n_dist']*rowb/ms/1e6:.0f} GB/s", flush=True)
    del eng, base
which = sys.argv[1:] or ["sift_u8", "gist", "deep"]
if "sift_u8" in which: run("SIFT1M-shaped u8", 1_000_000, 128, "u8", 0, 0.9, 200)
if "gist" in which: run("GIST1M-shaped f32 cosine", 1_000_000, 960, "f32", 1, 0.9, 200, "lowrank32")
if "gist" in which: run("GIST1M-shaped f32 L2", 1_000_000, 960, "f32", 0, 0.9, 200, "lowrank32")
if "deep" in which: run("DEEP100M/8 shard f32", 12_500_000, 96, "f32", 0, 0.9, 200)
if "sift1b" in which: run("SIFT1B/8 shard u8", 125_000_000, 128, "u8", 0, 0.9, 200)
