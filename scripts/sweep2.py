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
query = synthetic("lowrank16", 10_000, D, 4321, dev)
gt, _ = eng.bf_query(query, K)
eng.set_collect_counters(True)
for it in (64, 100, 128, 150, 200, 256, 300, 400):
    for tau in (0.8, 0.9, 1.0, 1.1, 1.25, 1.5, 2.0):
        for _ in range(2):
            ids, d = eng.query(query, K, tau, it)
        ms = min(eng.query(query, K, tau, it) and eng.last_timing_ms()["query_ms"] for _ in range(4))
        c = eng.last_query_counters()
        r = recall_at_k(ids, gt)
        if r > 0.985:
            print(f"it={it} tau={tau}: {ms:.2f} ms {10000/ms*1000:,.0f} qps recall={r:.4f} n_dist={c['n_dist']/1e4:.0f} n_pop={c['n_pop']/1e4:.0f}", flush=True)
