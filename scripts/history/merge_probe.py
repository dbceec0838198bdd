"""per-launch work counters of the merge kernel on a GPU-built 1M graph (exploration)"""
import sys, os, ctypes as C, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ggnn_amd as ggnn
from ggnn_amd import _lib
from ggnn_amd._lib import lib, check
from bench import synthetic
ggnn.set_log_level(-1)
dev = torch.device("cuda", 0)
N, D, K = 1_000_000, 128, 24
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
base = synthetic("lowrank16", N, D, 1234, dev)
if dt == "u8": base = base.to(torch.uint8)
eng = ggnn.GGNN(); eng.set_base_reference(base); eng.build(K, 0.5, 2)
view = _lib.GraphView(); check(lib().ggnn_get_graph(eng._h, 0, C.byref(view)))
cfg = view.config
for top, btm in ((1, 0), (2, 0), (3, 0), (3, 1), (3, 2)):
    Nb = cfg.Ns[btm]
    gb = torch.empty((Nb, K), dtype=torch.int32, device=dev); nn1 = torch.zeros(Nb, device=dev)
    nd = torch.zeros(Nb, dtype=torch.int32, device=dev)
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        check(lib().ggnn_op_merge(base.data_ptr(), 1 if dt == "u8" else 0, 0, cfg, view.graph, view.translation, view.selection,
                                  view.nn1_stats, 0.5, top, btm, gb.data_ptr(), nn1.data_ptr(), nd.data_ptr(), None))
        torch.cuda.synchronize(); ms = (time.perf_counter() - t) * 1e3
    tot = int(nd.sum().item())
    rowb = D * (1 if dt == "u8" else 4)
    print(f"merge {top}->{btm}: N={Nb} {ms:.1f} ms  n_dist/pt={tot/Nb:.0f}  {tot*rowb/ms/1e6:.0f} GB/s alg  {Nb/ms/1e3:.2f} Mpts/s", flush=True)
