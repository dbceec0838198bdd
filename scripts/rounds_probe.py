"""Query-kernel time against the NUMBER of queries of a launch, at one operating point: how much
of a 10 000-query batch is the partly filled second round of waves (one search per wave; 1024
SIMDs x 7 resident waves = 7168 searches per round).
    python scripts/rounds_probe.py <n_base> <dim> <kind> <tau:iters> [nq ...]   (kind: lowrank24 ...)
Prints one JSON line per batch size: kernel ms, queries/s, mean pops / evaluations per query."""
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

import ggnn_amd as ggnn
from bench import recall_at_k, synthetic

ggnn.set_log_level(-1)
n, d, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
tau, it = sys.argv[4].split(":")
tau, it = float(tau), int(it)
sizes = [int(x) for x in sys.argv[5:]] or [1024, 2048, 4096, 5000, 7168, 8192, 10000, 14336, 20000,
                                           100000]
dev = torch.device("cuda", 0)
base = synthetic(kind, n, d, 1234, dev)
big = synthetic(kind, max(sizes), d, 4321, dev)
eng = ggnn.GGNN()
eng.set_base_reference(base)
eng.set_return_results_on_gpu(True)
eng.build(24, 0.5, 2)
gt = eng.bf_query(big[:10_000].contiguous(), 10)[0]
print(json.dumps({"base": [n, d, kind], "tau": tau, "iters": it,
                  "build_s": eng.last_timing_ms()["build_ms"] / 1e3,
                  "recall_first_10k": round(recall_at_k(
                      eng.query(big[:10_000].contiguous(), 10, tau, it)[0], gt), 4)}), flush=True)
for nq in sizes:
    q = big[:nq].contiguous()
    eng.set_collect_counters(True)
    eng.query(q, 10, tau, it)
    cnt, rr = eng.last_query_counters(), eng.last_query_rows_read()
    eng.set_collect_counters(False)
    for _ in range(2):
        eng.query(q, 10, tau, it)
    ms = []
    for _ in range(5 if nq <= 20000 else 2):
        eng.query(q, 10, tau, it)
        ms.append(eng.last_timing_ms()["query_ms"])
    t = sorted(ms)[len(ms) // 2]
    print(json.dumps({"nq": nq, "ms": round(t, 4), "Mqps": round(nq / t / 1e3, 3),
                      "us_per_1000": round(t / nq * 1e6, 1), "counters": cnt, "rows": rr}),
          flush=True)
