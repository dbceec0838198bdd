"""event counts of bf_i8v2_kernel (debug build with -DGGNN_I8_STATS):
    make -C ggnn_amd/csrc TARGET=libggnn_dbg.so OBJDIR=build_dbg EXTRA=-DGGNN_I8_STATS
    GGNN_TEST_HOOKS=1 GGNN_AMD_LIB=ggnn_amd/csrc/libggnn_dbg.so python scripts/bf_i8_stats.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ggnn_amd import ops
from ggnn_amd._lib import lib
from bench import synthetic
dev = torch.device("cuda", 0)
base = synthetic("lowrank16", 1_000_000, 128, 1234, dev).to(torch.uint8)
query = synthetic("lowrank16", 10_000, 128, 4321, dev).to(torch.uint8)
out = (C.c_ulonglong * 16)()
ops.bf_query(base, query, 10); torch.cuda.synchronize()
lib().ggnn_debug_i8_stats(out, 1)
ops.bf_query(base, query, 10); torch.cuda.synchronize()
lib().ggnn_debug_i8_stats(out, 1)
names = ["tile-sets entered", "filter hits (lane 0)", "appended (lane 0)", "flush calls",
         "flush iterations", "tile-sets", "offset refreshes", "-", "cycles in flush",
         "cycles in refresh_offsets", "cycles in tile_set (incl. its flushes)", "cycles in main loop (per wave, summed)",
         "cycles in refresh (exchange)", "cycles MFMA..max3..any per tile", "cycles stage_store (incl. wait for the loads)", "cycles in barrier"]
ts = max(1, out[5])
for n, v in zip(names, out):
    print(f"{n:20s} {v:12d}  per tile-set {v / ts:.4f}")
