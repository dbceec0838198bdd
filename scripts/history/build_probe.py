"""Graph build time per XCD_MAP setting + the merge / sym work counters (ggnn_last_build_work).
    python scripts/build_probe.py [D] [N]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ggnn_amd as ggnn
from ggnn_amd import _lib
from bench import synthetic
ggnn.set_log_level(-1)
dev = torch.device("cuda", 0)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
base = synthetic("lowrank16", N, D, 1234, dev)
out = {}
for xm in (0, 1, 3, 0, 3):
    with _lib.hooks(XCD_MAP=xm):
        eng = ggnn.GGNN(); eng.set_base_reference(base); eng.build(24, 0.5, 2)
        out.setdefault(f"xcd_map={xm}", []).append(round(eng.last_timing_ms()["build_ms"] / 1e3, 4))
        del eng
for xm in (0, 3):
    with _lib.hooks(XCD_MAP=xm):
        eng = ggnn.GGNN(); eng.set_base_reference(base); eng.set_collect_counters(True)
        eng.build(24, 0.5, 2)
        out[f"work xcd_map={xm}"] = eng.last_build_work()
        del eng
print(json.dumps(out))
