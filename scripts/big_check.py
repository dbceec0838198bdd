import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ggnn_amd as ggnn
from bench import synthetic, recall_at_k
ggnn.set_log_level(-1)
dev = torch.device("cuda", 0)
N = int(sys.argv[1]); D = 128
dt = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else torch.uint8
base = torch.empty((N, D), dtype=dt, device=dev)
for lo in range(0, N, 5_000_000):
    hi = min(N, lo + 5_000_000)
    base[lo:hi] = synthetic("lowrank16", hi - lo, D, 1234 + lo, dev).to(dt)
query = synthetic("lowrank16", 10_000, D, 4321, dev).to(dt)
eng = ggnn.GGNN(); eng.set_base_reference(base); eng.set_return_results_on_gpu(True)
t = time.time(); eng.build(24, 0.5, 2); print("build", time.time() - t, flush=True)
gt, gtd = eng.bf_query(query, 10)
print("gt id range", gt.min().item(), gt.max().item(), "dist0 mean", gtd[:, 0].float().mean().item(), flush=True)
eng.set_collect_counters(True)
points = ((0.9, 200), (1.0, 400), (1.5, 400), (1.5, 1000), (2.0, 2000))
if dt == torch.float32:
    points = points[:3]
for tau, it in points:
    ids, d = eng.query(query, 10, tau, it)
    if dt == torch.float32:  # the pre-screen must not change anything, also beyond 2^32 bytes of codes
        eng.set_prescreen(False); ids2, d2 = eng.query(query, 10, tau, it); eng.set_prescreen(True)
        assert torch.equal(ids, ids2) and torch.equal(d, d2)
        ids, d = eng.query(query, 10, tau, it)
    c = eng.last_query_counters()
    print(f"tau={tau} it={it}: {eng.last_timing_ms()['query_ms']:.2f} ms recall={recall_at_k(ids, gt):.4f} c1={(ids[:,0]==gt[:,0]).float().mean().item():.4f} pops/q={c['n_pop']/1e4:.0f} id range {ids.min().item()} {ids.max().item()} d0 {d[:,0].mean().item():.1f}", flush=True)
import ctypes as C
from ggnn_amd import _lib
v = _lib.GraphView(); _lib.check(_lib.lib().ggnn_get_graph(eng._h, 0, C.byref(v)))
c = v.config
print({k: getattr(c, k) for k in ("N", "G", "S", "S0", "S0_off", "SG", "SG_off", "N_all", "ST_all")}, list(c.Ns))
