"""Long searches (1000-2000 iterations): tag set vs ring scan on the harder synthetic bases.
    python scripts/longring_probe.py [dataset ...]"""
import os, sys, json
sys.path.insert(0, os.getcwd())
import torch
import ggnn_amd as ggnn
from ggnn_amd import _lib
from bench import synthetic, recall_at_k
ggnn.set_log_level(-1)
dev = torch.device("cuda", 0)
kinds = sys.argv[1:] or ["lowrank24", "lowrank32"]
out = {}
for kind in kinds:
    base = synthetic(kind, 1_000_000, 128, 1234, dev)
    q = synthetic(kind, 10_000, 128, 4321, dev)
    eng = ggnn.GGNN(); eng.set_base_reference(base); eng.set_return_results_on_gpu(True)
    eng.build(24, 0.5, 2)
    gt, _ = eng.bf_query(q, 10)
    r = {}
    for tau, it in ((1.0, 600), (1.0, 800), (1.0, 1000), (1.2, 1000), (1.0, 1500), (1.0, 2000)):
        row = {}
        for tag in (1, 0):
            with _lib.hooks(VIS_TAG_SET=tag):
                for _ in range(2): eng.query(q, 10, tau, it)
                ms = []
                for _ in range(3):
                    ids, d = eng.query(q, 10, tau, it); ms.append(eng.last_timing_ms()["query_ms"])
            row["tag" if tag else "scan"] = round(sum(ms) / len(ms), 3)
            row["recall"] = round(recall_at_k(ids, gt), 4)
            row["sum" + str(tag)] = int(ids.sum())
        row["identical"] = row.pop("sum1") == row.pop("sum0")
        row["qps_tag"] = round(10_000 / row["tag"] * 1e3)
        r[f"tau{tau}_it{it}"] = row
    out[kind] = r
    del eng, base, q
    torch.cuda.empty_cache()
print(json.dumps(out))
