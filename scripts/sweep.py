"""tau / batch-size sweep of the query kernel on the bench workload (exploration)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ggnn_amd as ggnn
from bench import synthetic, recall_at_k
ggnn.set_log_level(-1)
N, D, K = 1_000_000, 128, 10
dev = torch.device("cuda", 0)
base = synthetic("lowrank16", N, D, 1234, dev)
eng = ggnn.GGNN(); eng.set_base(base); eng.set_return_results_on_gpu(True); eng.build(24, 0.5, 2)
for nq in (10_000, 100_000):
    query = synthetic("lowrank16", nq, D, 4321, dev)
    gt, _ = eng.bf_query(query[:10_000], K)
    for tau, it in ((0.8, 400), (0.85, 400), (0.9, 400), (0.95, 400), (1.0, 400), (1.0, 200), (1.2, 400)):
        for _ in range(2):
            ids, d = eng.query(query, K, tau, it)
        ms = min(eng.query(query, K, tau, it) and eng.last_timing_ms()["query_ms"] for _ in range(5))
        print(f"nq={nq} tau={tau} it={it}: {ms:.2f} ms  {nq/ms*1000:,.0f} qps  recall@10={recall_at_k(ids[:10000], gt):.4f}", flush=True)
