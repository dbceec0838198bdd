"""bf_query timing for uint8 rows: bf_time_u8.py [n=1M] [k=10] [D=128]; GGNN_TEST_HOOKS=1 GGNN_BF_I8_V1=1 times the
LDS-list kernel"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ggnn_amd import ops
from bench import synthetic
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
D = int(sys.argv[3]) if len(sys.argv) > 3 else 128
base = synthetic("lowrank16", n, D, 1234, dev).to(torch.uint8)
query = synthetic("lowrank16", 10_000, D, 4321, dev).to(torch.uint8)
for _ in range(2):
    ids, d = ops.bf_query(base, query, k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ids, d = ops.bf_query(base, query, k)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
tag = "v1 (LDS lists)" if os.environ.get("GGNN_BF_I8_V1") else "v2 (register sets)"
print(f"bf_query u8 {n}x{D} k={k} {tag}: {ms:.2f} ms  {2*1e4*n*D/ms/1e9:.1f} Top/s "
      f"checksum {int(ids.sum())} {float(d.sum())}")
