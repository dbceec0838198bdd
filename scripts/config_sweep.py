"""operating points (recall >= 0.99) for the secondary BASELINE configs (exploration)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ggnn_amd as ggnn
from bench import synthetic, recall_at_k
ggnn.set_log_level(-1)
dev = torch.device("cuda", 0)
def run(name, N, D, measure, kind, points):
    base = synthetic(kind, N, D, 1234, dev); query = synthetic(kind, 10_000, D, 4321, dev)
    eng = ggnn.GGNN(); eng.set_base_reference(base); eng.set_return_results_on_gpu(True)
    t = time.time(); eng.build(24, 0.5, 2, measure); tb = time.time() - t
    gt, _ = eng.bf_query(query, 10, measure); bf = eng.last_timing_ms()["bf_query_ms"]
    print(f"{name}: build {tb:.2f}s bf {bf:.0f} ms", flush=True)
    for tau, it in points:
        for _ in range(2): ids, d = eng.query(query, 10, tau, it, measure)
        ms = eng.last_timing_ms()["query_ms"]
        print(f"   tau={tau} it={it}: {ms:.2f} ms {1e7/ms:,.0f} qps recall@10={recall_at_k(ids, gt):.4f}", flush=True)
pts = [(0.9, 200), (1.0, 400), (1.2, 400), (1.5, 400), (1.5, 800), (2.0, 1000)]
which = sys.argv[1:] or ["gist", "deep"]
if "gist" in which:
    run("GIST1M-shaped f32 cosine (lowrank32)", 1_000_000, 960, 1, "lowrank32", pts)
    run("GIST1M-shaped f32 cosine (lowrank16)", 1_000_000, 960, 1, "lowrank16", pts[:3])
if "deep" in which:
    run("DEEP100M/8 shard f32 L2", 12_500_000, 96, 0, "lowrank16", pts[:4])
