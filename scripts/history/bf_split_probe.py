"""Equal tile ranges vs aligned slices in the single-chunk f32 bf kernel, for several Nq
(hook BF_SLICES = s gives ranges of 1/s of a query block: the aligned slices of rounds 1-3)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from ggnn_amd import _lib, ops
from bench import synthetic
dev = torch.device("cuda", 0)
base = synthetic("lowrank16", 1_000_000, 128, 1234, dev)
def t(q, **hooks):
    with _lib.hooks(**hooks):
        for _ in range(2):
            r = ops.bf_query(base, q, 10)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            r = ops.bf_query(base, q, 10)
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 3, int(r[0].sum())
for nq in (2_000, 10_000, 30_000, 100_000):
    q = synthetic("lowrank16", nq, 128, 4321, dev)
    qblocks = (nq + 127) // 128
    aligned = max(1, min(32, 768 // qblocks))
    a, ca = t(q)
    b, cb = t(q, BF_SLICES=aligned)
    fl = 2 * nq * 1e6 * 128
    print(f"Nq {nq}: equal ranges {a:.2f} ms ({fl / a / 1e9 / 157.3:.3f})   {aligned} aligned slices "
          f"{b:.2f} ms ({fl / b / 1e9 / 157.3:.3f})   same ids {ca == cb}")
