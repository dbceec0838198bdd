import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ggnn_amd as ggnn
from bench import synthetic, recall_at_k
ggnn.set_log_level(-1)
dev = torch.device("cuda", 0)
N, D = 1_000_000, 128
base = synthetic("lowrank16", N, D, 1234, dev).to(torch.uint8)
query = synthetic("lowrank16", 100_000, D, 4321, dev).to(torch.uint8)
eng = ggnn.GGNN(); eng.set_base_reference(base); eng.set_return_results_on_gpu(True); eng.build(24, 0.5, 2)
for nq in (10_000, 100_000):
    for _ in range(3):
        eng.query(query[:nq].contiguous(), 10, 0.9, 200)
    print(nq, eng.last_timing_ms()["query_ms"], flush=True)
