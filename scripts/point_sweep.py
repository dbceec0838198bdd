"""Fine sweep of an operating point (tau_query x max_iterations) on one synthetic 1M x 128 base:
kernel time and recall@10 on the tuning query set (seed 4321) and on two held-out sets.
    python scripts/point_sweep.py [kind] [taus] [iterations]
    python scripts/point_sweep.py lowrank24 0.9,1.0,1.1 600,650,700,750,800"""
import os, sys, json
sys.path.insert(0, os.getcwd())
import torch
import ggnn_amd as ggnn
from bench import synthetic, recall_at_k
ggnn.set_log_level(-1)
kind = sys.argv[1] if len(sys.argv) > 1 else "lowrank16"
taus = [float(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (0.8, 0.85, 0.9, 0.95, 1.0, 1.1, 1.25)
its = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else (110, 120, 130, 140, 150, 160, 175)
floor = float(sys.argv[4]) if len(sys.argv) > 4 else 0.9905
dev = torch.device("cuda", 0)
base = synthetic(kind, 1_000_000, 128, 1234, dev)
qs = {"tune": synthetic(kind, 10_000, 128, 4321, dev), "held": synthetic(kind, 10_000, 128, 8642, dev),
      "held2": synthetic(kind, 10_000, 128, 1357, dev)}
eng = ggnn.GGNN(); eng.set_base_reference(base); eng.set_return_results_on_gpu(True)
eng.build(24, 0.5, 2)
gts = {k: eng.bf_query(q, 10)[0] for k, q in qs.items()}
rows = []
for tau in taus:
    for it in its:
        ms = []
        for _ in range(2): eng.query(qs["tune"], 10, tau, it)
        for _ in range(3):
            eng.query(qs["tune"], 10, tau, it); ms.append(eng.last_timing_ms()["query_ms"])
        rec = {k: round(recall_at_k(eng.query(q, 10, tau, it)[0], gts[k]), 4) for k, q in qs.items()}
        rows.append({"kind": kind, "tau": tau, "it": it, "ms": round(sum(ms) / len(ms), 4), **rec})
        print(json.dumps(rows[-1]), flush=True)
ok = [r for r in rows if min(r["tune"], r["held"], r["held2"]) >= floor]
print(f"fastest with recall >= {floor} on all three sets:", json.dumps(sorted(ok, key=lambda r: r["ms"])[:5]))
