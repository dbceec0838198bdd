"""fine sweep of the query operating point around tau 0.9 / 200 iterations (SIFT1M-shaped f32)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ggnn_amd as ggnn
from bench import synthetic, recall_at_k
ggnn.set_log_level(-1)
dev = torch.device("cuda", 0)
base = synthetic("lowrank16", 1_000_000, 128, 1234, dev); q = synthetic("lowrank16", 10_000, 128, 4321, dev)
for trial in range(2):
    eng = ggnn.GGNN(); eng.set_base_reference(base); eng.set_return_results_on_gpu(True); eng.build(24, 0.5, 2)
    gt, _ = eng.bf_query(q, 10)
    for tau in (0.75, 0.8, 0.85, 0.9):
        for it in (128, 150, 175, 200):
            for _ in range(3): ids, d = eng.query(q, 10, tau, it)
            ms = eng.last_timing_ms()["query_ms"]
            print(f"build {trial} tau={tau} it={it}: {ms:.3f} ms {1e7/ms:,.0f} q/s recall {recall_at_k(ids, gt):.4f}", flush=True)
