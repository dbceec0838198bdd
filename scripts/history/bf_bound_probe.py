"""Upper limit of a bound shared by the slices of a query (stats build, see bf_phase_cycles.py):
the lists of the f32 bf kernel start from a per-query bound just above the true 18th-best distance,
i.e. only true candidates are ever inserted."""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from ggnn_amd import _lib, ops
from bench import synthetic
dev = torch.device("cuda", 0)
base = synthetic("lowrank16", 1_000_000, 128, 1234, dev)
q = synthetic("lowrank16", 10_000, 128, 4321, dev)
h = C.CDLL(_lib.LIB_PATH)
def t(tag):
    for _ in range(2):
        r = ops.bf_query(base, q, 10)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        r = ops.bf_query(base, q, 10)
    e1.record(); torch.cuda.synchronize()
    print(f"{tag}: {e0.elapsed_time(e1) / 5:.2f} ms  checksum {int(r[0].sum())} {float(r[1].sum()):.1f}")
t("lists start at +inf")
ids, d = ops.bf_query(base, q, 18)
factor = float(sys.argv[1]) if len(sys.argv) > 1 else 1.01
bound = (d[:, 17] * factor + 1.0).contiguous().float()
h.ggnn_debug_bf_bound(C.c_void_p(bound.data_ptr()))
t(f"lists start at {factor} x the 18th-best distance")
