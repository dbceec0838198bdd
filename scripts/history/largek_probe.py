"""query time vs KQuery (register-resident list R = 1..32 up to K = 2031, LDS-resident list above)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ggnn_amd as ggnn
from bench import synthetic, recall_at_k
ggnn.set_log_level(-1)
dev = torch.device("cuda", 0)
base = synthetic("lowrank16", 1_000_000, 128, 1234, dev)
query = synthetic("lowrank16", 2_000, 128, 4321, dev)
eng = ggnn.GGNN(); eng.set_base_reference(base); eng.set_return_results_on_gpu(True)
eng.build(24, 0.5, 2)
for K, it in ((10, 400), (100, 1000), (200, 1000), (239, 1000), (240, 1000), (300, 1000), (500, 2000), (1000, 4000)):
    gt, _ = eng.bf_query(query, min(K, 100))
    for _ in range(2):
        ids, d = eng.query(query, K, 0.9, it)
    ms = eng.last_timing_ms()["query_ms"]
    print(f"K={K} it={it}: {ms:.2f} ms  {2000/ms*1e3:,.0f} q/s  recall@{min(K,100)}={recall_at_k(ids[:, :min(K,100)].contiguous(), gt):.4f}", flush=True)
