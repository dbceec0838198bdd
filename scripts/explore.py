"""Exploration on the GPU box: build/query time and recall at SIFT1M scale for several synthetic
distributions.  Not part of the product."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ggnn_amd as ggnn

def gen(kind, N, D, seed, dev="cuda"):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    if kind == "iid":
        return torch.randint(0, 256, (N, D), generator=g, device=dev).float()
    if kind.startswith("lowrank"):
        L = int(kind[7:] or 16)
        ga = torch.Generator(device=dev); ga.manual_seed(777)
        A = torch.randn(L, D, generator=ga, device=dev) * (40.0 / L ** 0.5)
        z = torch.randn(N, L, generator=g, device=dev)
        return (128 + z @ A).round().clamp(0, 255)
    raise ValueError(kind)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ["lowrank16", "iid"]
Nq, K, D = 10_000, 10, 128
for kind in kinds:
    base = gen(kind, N, D, 1234); query = gen(kind, Nq, D, 4321)
    eng = ggnn.GGNN(); eng.set_base(base); eng.set_return_results_on_gpu(True)
    t = time.time(); eng.build(24, 0.5, 2); torch.cuda.synchronize(); tb = time.time() - t
    print(f"[{kind}] N={N} build wall {tb:.2f}s engine {eng.last_timing_ms()['build_ms']/1000:.2f}s", flush=True)
    t = time.time(); gt, _ = eng.bf_query(query, K); print(f"  bf_query {eng.last_timing_ms()['bf_query_ms']:.1f} ms (wall {time.time()-t:.2f}s)", flush=True)
    eng.set_collect_counters(True)
    for tau, it in ((0.34, 200), (0.41, 200), (0.51, 200), (0.64, 400), (0.8, 400), (1.0, 400)):
        ids, d = eng.query(query, K, tau, it)
        ms = eng.last_timing_ms()["query_ms"]; c = eng.last_query_counters()
        inter = (ids.unsqueeze(2) == gt.unsqueeze(1)).any(2).float().mean().item()
        c1 = (ids[:, 0] == gt[:, 0]).float().mean().item()
        bytes_q = D * 4 * Nq + c["n_dist"] * D * 4 + c["n_pop"] * 24 * 4 + Nq * (32 * 4 + 8 + K * 8)
        print(f"  tau={tau} it={it}: {ms:.2f} ms -> {Nq/ms*1000:,.0f} qps  recall@10={inter:.4f} c@1={c1:.4f} n_dist/q={c['n_dist']/Nq:.0f} n_pop/q={c['n_pop']/Nq:.1f}  alg GB/s={bytes_q/ms/1e6:.0f}", flush=True)
    del eng
