import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = '''
import sys, os
sys.path.insert(0, %r)
import torch
from ggnn_amd import ops
from bench import synthetic
dev = torch.device("cuda", 0)
base = synthetic("lowrank16", 1_000_000, 128, 1234, dev); q = synthetic("lowrank16", 10_000, 128, 4321, dev)
for _ in range(2): ops.bf_query(base, q, 10)
torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(3): ops.bf_query(base, q, 10)
e.record(); torch.cuda.synchronize(); print(os.environ.get("GGNN_BF_SLICES"), s.elapsed_time(e) / 3, "ms")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sl in (6, 10, 12, 13, 16, 19, 20, 26, 32, 45):
    env = dict(os.environ, GGNN_BF_SLICES=str(sl))
    print(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip(), flush=True)
