"""Shader cycles per phase of a pop of the headline query kernel (stats build):
    make -C ggnn_amd/csrc OBJDIR=build_ph TARGET=libggnn_ph.so EXTRA=-DGGNN_PHASE_CYCLES   (query.o only)
    GGNN_TEST_HOOKS=1 GGNN_AMD_LIB=$PWD/ggnn_amd/csrc/libggnn_ph.so python scripts/phase_cycles.py
Reports, for batches of 1024 (one wave per SIMD: a lone wave's latency chain) and 10 000 queries,
the average cycles per pop spent in each phase (every phase boundary drains the memory counters)."""
import ctypes as C, os, sys, json
sys.path.insert(0, os.getcwd())
import torch
import ggnn_amd as ggnn
from ggnn_amd import _lib
from bench import synthetic
ggnn.set_log_level(-1)
dev = torch.device("cuda", 0)
base = synthetic("lowrank16", 1_000_000, 128, 1234, dev)
q = synthetic("lowrank16", 10_000, 128, 4321, dev)
eng = ggnn.GGNN(); eng.set_base_reference(base); eng.set_return_results_on_gpu(True)
eng.build(24, 0.5, 2)
h = C.CDLL(_lib.LIB_PATH)
names = ["pop", "graph row wait", "filter", "compaction (+ wait for the speculative row)",
         "code rows: issue + wait", "verdicts + compaction", "float rows: issue + wait",
         "distances", "replay (pushes)"]
out = {}
for nq in (1024, 10_000):
    x = q[:nq].contiguous()
    for _ in range(2):
        eng.query(x, 10, 0.9, 175)
    acc = (C.c_ulonglong * 16)()
    h.ggnn_debug_phase_cycles(acc, 1)
    eng.set_collect_counters(True)
    eng.query(x, 10, 0.9, 175)
    pops = eng.last_query_counters()["n_pop"]
    ms = eng.last_timing_ms()["query_ms"]
    eng.set_collect_counters(False)
    h.ggnn_debug_phase_cycles(acc, 1)
    per = {names[i]: round(acc[i] / pops, 1) for i in range(len(names))}
    per["total per pop"] = round(sum(acc[i] for i in range(len(names))) / pops, 1)
    per["pops per query"] = round(pops / nq, 1)
    per["kernel ms (stats build)"] = round(ms, 3)
    out[f"{nq} queries"] = per
print(json.dumps(out, indent=1))
