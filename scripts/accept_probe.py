import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ggnn_amd as ggnn
from oracle import oracle as orc
from bench import synthetic
ggnn.set_log_level(-1)
dev = torch.device("cuda", 0)
base = synthetic("lowrank16", 1_000_000, 128, 1234, dev); q = synthetic("lowrank16", 400, 128, 4321, dev)
eng = ggnn.GGNN(); eng.set_base_reference(base); eng.build(24, 0.5, 2)
g = eng.get_graph(0)
lib = orc.lib(); lib.orc_accept_total.restype = C.c_uint64
orc.set_fast_distance(True)
for tau, it in ((0.9, 200), (1.0, 400)):
    lib.orc_accept_total(1)
    ids, d, nd, npop = orc.query(base.cpu().numpy(), q.cpu().numpy(), g.graph[0].view.numpy(), g.translation[3].view.numpy().reshape(-1),
                                 g.nn1_stats.view.numpy().reshape(-1), 10, tau, it, counters=True)
    acc = lib.orc_accept_total(0)
    print(f"tau={tau} it={it}: n_dist/q={nd.mean():.0f} n_pop/q={npop.mean():.0f} accepted/q={acc/400:.0f} accept fraction={acc/nd.sum():.3f}")
