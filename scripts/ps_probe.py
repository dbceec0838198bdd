"""pre-screen on/off: query kernel time on one graph for several shapes (exploration)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ggnn_amd as ggnn
from bench import synthetic, recall_at_k
ggnn.set_log_level(-1)
dev = torch.device("cuda", 0)
def run(name, N, D, kind, nq, points, measure=0):
    base = synthetic(kind, N, D, 1234, dev); query = synthetic(kind, nq, D, 4321, dev)
    eng = ggnn.GGNN(); eng.set_base_reference(base); eng.set_return_results_on_gpu(True)
    eng.build(24, 0.5, 2, measure)
    print(f"{name}: build {eng.last_timing_ms()['build_ms']:.0f} ms", flush=True)
    gt, _ = eng.bf_query(query[:2000], 10, measure)
    res = {}
    for tau, it in points:
        for on in (True, False):
            eng.set_prescreen(on)
            for _ in range(3): ids, d = eng.query(query, 10, tau, it, measure)
            ms = eng.last_timing_ms()["query_ms"]
            res[on] = (ids.clone(), d.clone())
            print(f"{name} nq={nq} tau={tau} it={it} prescreen={on}: {ms:.3f} ms {nq/ms*1e3:,.0f} qps recall {recall_at_k(ids[:2000], gt):.4f}", flush=True)
        assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
which = sys.argv[1:] or ["sift", "sift100k", "gauss", "d256", "gist"]
if "sift" in which: run("sift-shaped", 1_000_000, 128, "lowrank16", 10_000, [(0.9, 200), (1.0, 400)])
if "sift100k" in which: run("sift-shaped", 1_000_000, 128, "lowrank16", 100_000, [(0.9, 200)])
if "gauss" in which: run("fractional", 1_000_000, 128, "lowrankf16", 10_000, [(0.9, 200)])
if "d256" in which: run("d256", 500_000, 256, "lowrank16", 10_000, [(0.9, 200)])
if "gist" in which: run("gist-shaped L2", 500_000, 960, "lowrank32", 10_000, [(0.9, 200)])
if "gistcos" in which: run("gist-shaped cosine", 1_000_000, 960, "lowrank32", 10_000, [(0.9, 200), (1.5, 400)], 1)
