"""query latency for small batches (SIFT1M-shaped, k=10, tau 0.9 / 200 iterations)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ggnn_amd as ggnn
from bench import synthetic
ggnn.set_log_level(-1)
dev = torch.device("cuda", 0)
base = synthetic("lowrank16", 1_000_000, 128, 1234, dev)
eng = ggnn.GGNN(); eng.set_base_reference(base); eng.set_return_results_on_gpu(True); eng.build(24, 0.5, 2)
for nq in (1, 16, 256, 1024, 4096):
    q = synthetic("lowrank16", nq, 128, 4321, dev)
    for _ in range(5): eng.query(q, 10, 0.9, 200)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): eng.query(q, 10, 0.9, 200)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t) / 50 * 1e3
    print(f"nq={nq}: wall {wall:.3f} ms per call, kernel {eng.last_timing_ms()['query_ms']:.3f} ms, {nq / wall * 1e3:,.0f} q/s")
