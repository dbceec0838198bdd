import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ggnn_amd as ggnn
from bench import synthetic, recall_at_k
ggnn.set_log_level(-1)
dev = torch.device("cuda", 0)
N, D = 1_000_000, 128
for dt in ("f32", "u8"):
    base = synthetic("lowrank16", N, D, 1234, dev); query = synthetic("lowrank16", 100_000, D, 4321, dev)
    if dt == "u8": base, query = base.to(torch.uint8), query.to(torch.uint8)
    eng = ggnn.GGNN(); eng.set_base_reference(base); eng.set_return_results_on_gpu(True); eng.build(24, 0.5, 2)
    for nq in (1000, 10_000, 100_000):
        q = query[:nq].contiguous()
        ms = min(eng.query(q, 10, 0.9, 200) and eng.last_timing_ms()["query_ms"] for _ in range(6))
        print(f"{dt} nq={nq}: {ms:.3f} ms  {nq/ms*1e3:,.0f} qps", flush=True)
    del eng
