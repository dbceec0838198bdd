"""Operating point of the north star's 100M x 96 base (8 shards x 12.5M, BASELINE configs[3]) by
MERGED recall: all 8 shards resident on ONE GPU (the `one_gpu_same_base` point of bench.py
--gpus N), exact ground truth over the whole base, blocking 10k-query steps per (tau, iterations)
on the tuning query set and a held-out set.
    python scripts/shard8_probe.py [n_shard] [dim] [tau:iters ...]"""
import json
import os
import sys
import time
import types

sys.path.insert(0, os.getcwd())
import torch

import ggnn_amd as ggnn
from bench import big_base, recall_at_k, synthetic

ggnn.set_log_level(-1)
n_shard = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 96
points = [tuple(float(x) for x in p.split(":")) for p in sys.argv[3:]] or [(1.0, 400)]
dev = torch.device("cuda", 0)
args = types.SimpleNamespace(n_base=n_shard, dim=dim, dataset="lowrank16")
base = big_base(args, 8 * n_shard, 0, dev)
qs = {"tune": synthetic("lowrank16", 10_000, dim, 4321, dev),
      "held": synthetic("lowrank16", 10_000, dim, 8642, dev)}
eng = ggnn.GGNN()
eng.set_base_reference(base)
eng.set_shard_size(n_shard)
eng.set_return_results_on_gpu(False)
t0 = time.perf_counter()
eng.build(24, 0.5, 2)
print(f"8 x {n_shard} x {dim}: build {time.perf_counter() - t0:.1f} s", flush=True)
# exact ground truth over the whole base: brute force per shard + merge (bf_query of a handle
# works on its whole base)
gts = {}
for k, q in qs.items():
    gts[k] = eng.bf_query(q, 10)[0].to(dev)
for tau, it in points:
    it = int(it)
    for _ in range(2):
        eng.query(qs["tune"], 10, tau, it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ids, _ = eng.query(qs["tune"], 10, tau, it)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    row = {"tau": tau, "it": it, "ms_blocking_10k": round(ms, 3),
           "qps": round(10_000 / ms * 1e3), "kernel_ms_sum": round(eng.last_timing_ms()["query_ms"], 3),
           "recall": {k: round(recall_at_k(eng.query(q, 10, tau, it)[0].to(dev), gts[k]), 4)
                      for k, q in qs.items()}}
    print(json.dumps(row), flush=True)
