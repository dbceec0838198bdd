"""bf_query timing (HIP events over ops.bf_query, 10k x 1M x 128 f32, k=10)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ggnn_amd import ops
from bench import synthetic
dev = torch.device("cuda", 0)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
base = synthetic("lowrank16", 1_000_000, D, 1234, dev)
query = synthetic("lowrank16", 10_000, D, 4321, dev)
for _ in range(2):
    ids, d = ops.bf_query(base, query, 10)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ids, d = ops.bf_query(base, query, 10)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"bf_query D={D}: {ms:.2f} ms  {2*1e4*1e6*D/ms/1e9:.1f} TFLOP/s  frac {2*1e4*1e6*D/ms/1e9/157.3:.3f}")
