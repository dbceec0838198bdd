"""Fine sweep of the headline operating point (tau_query x max_iterations) on the default base:
kernel time and recall@10 on the tuning query set (seed 4321) and on a held-out set (seed 8642).
    python scripts/point_sweep.py"""
import os, sys, json
sys.path.insert(0, os.getcwd())
import torch
import ggnn_amd as ggnn
from bench import synthetic, recall_at_k
ggnn.set_log_level(-1)
dev = torch.device("cuda", 0)
base = synthetic("lowrank16", 1_000_000, 128, 1234, dev)
qs = {"tune": synthetic("lowrank16", 10_000, 128, 4321, dev), "held": synthetic("lowrank16", 10_000, 128, 8642, dev),
      "held2": synthetic("lowrank16", 10_000, 128, 1357, dev)}
eng = ggnn.GGNN(); eng.set_base_reference(base); eng.set_return_results_on_gpu(True)
eng.build(24, 0.5, 2)
gts = {k: eng.bf_query(q, 10)[0] for k, q in qs.items()}
rows = []
for tau in (0.8, 0.85, 0.9, 0.95, 1.0, 1.1, 1.25):
    for it in (110, 120, 130, 140, 150, 160, 175):
        ms = []
        for _ in range(2): eng.query(qs["tune"], 10, tau, it)
        for _ in range(5):
            eng.query(qs["tune"], 10, tau, it); ms.append(eng.last_timing_ms()["query_ms"])
        rec = {k: round(recall_at_k(eng.query(q, 10, tau, it)[0], gts[k]), 4) for k, q in qs.items()}
        rows.append({"tau": tau, "it": it, "ms": round(sum(ms) / len(ms), 4), **rec})
        print(rows[-1], flush=True)
ok = [r for r in rows if min(r["tune"], r["held"], r["held2"]) >= 0.9905]
print("fastest with recall >= 0.9905 on all three sets:", sorted(ok, key=lambda r: r["ms"])[:5])
