"""Shader cycles per phase of a tile of the single-chunk f32 bf kernel (stats build):
    make -C ggnn_amd/csrc OBJDIR=build_bfph TARGET=libggnn_bfph.so EXTRA=-DGGNN_BF_PHASE
    GGNN_TEST_HOOKS=1 GGNN_AMD_LIB=$PWD/ggnn_amd/csrc/libggnn_bfph.so python scripts/bf_phase_cycles.py
Per wave and tile: MFMA chain + test of the previous tile | insertions | stage store (waits for the
prefetched rows) | barrier.  A tile's 64 v_mfma_f32_32x32x2_f32 occupy the matrix pipe for 4096
cycles; two waves share a SIMD."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.getcwd())
import torch
from ggnn_amd import _lib, ops
from bench import synthetic
dev = torch.device("cuda", 0)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 128   # > 128: the chunked kernel, per (tile, chunk) pair
base = synthetic("lowrank16", 1_000_000, D, 1234, dev)
q = synthetic("lowrank16", 10_000, D, 4321, dev)
h = C.CDLL(_lib.LIB_PATH)
for _ in range(2):
    ops.bf_query(base, q, 10)
torch.cuda.synchronize()
acc = (C.c_ulonglong * 8)()
h.ggnn_debug_bf_phase(acc, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.bf_query(base, q, 10); e1.record(); torch.cuda.synchronize()
h.ggnn_debug_bf_phase(acc, 1)
names = ["mfma chain + test", "insertions", "stage store", "barrier"]
tiles = acc[4]
out = {n: round(acc[i] / tiles, 1) for i, n in enumerate(names)}
out["total per wave-tile"] = round(sum(acc[i] for i in range(4)) / tiles, 1)
out["wave-tiles"] = tiles
out["bf_query ms (stats build)"] = round(e0.elapsed_time(e1), 2)
print(json.dumps(out, indent=1))
