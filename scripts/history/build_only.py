"""graph build only (for rocprofv3 --stats): python scripts/build_only.py D [cos]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ggnn_amd as ggnn
from bench import synthetic
ggnn.set_log_level(-1)
dev = torch.device("cuda", 0)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
measure = 1 if len(sys.argv) > 2 and sys.argv[2] == "cos" else 0
base = synthetic("lowrank16", 1_000_000, D, 1234, dev)
for rep in range(2):
    eng = ggnn.GGNN(); eng.set_base_reference(base); eng.build(24, 0.5, 2, measure)
    print("build_s", eng.last_timing_ms()["build_ms"] / 1e3, flush=True)
    del eng
