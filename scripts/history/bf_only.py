import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ggnn_amd as ggnn
from ggnn_amd import ops
from bench import synthetic
dev = torch.device("cuda", 0)
base = synthetic("lowrank16", 1_000_000, 128, 1234, dev)
query = synthetic("lowrank16", 10_000, 128, 4321, dev)
for _ in range(3):
    ids, d = ops.bf_query(base, query, 10)
torch.cuda.synchronize()
