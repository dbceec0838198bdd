"""cross-check of the brute-force paths at scale: matrix-core kernels vs the scan kernel"""
import os, sys, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ggnn_amd import ops
from bench import synthetic
dev = torch.device("cuda", 0)
mode = sys.argv[1]
dt = torch.uint8 if mode.startswith("u8") else torch.float32
base = synthetic("lowrank16", 1_000_000, 128, 1234, dev).to(dt)
query = synthetic("lowrank16", 2048, 128, 4321, dev).to(dt)
ids, d = ops.bf_query(base, query, 10)
torch.save((ids.cpu(), d.cpu()), sys.argv[2])
print(mode, "done", ids[0, :4].tolist())
