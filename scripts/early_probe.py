"""Query-kernel order A/B (hook QUERY_EARLY = 0 | 1, traversal.hpp "Early rows") on one shape:
identical ids / distances / counters required, kernel time per operating point, recall@10.
    python scripts/early_probe.py <n_base> <dim> <f32|u8> [tau:iters ...]
Run against experimental builds with GGNN_TEST_HOOKS=1 GGNN_AMD_LIB=..."""
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

import ggnn_amd as ggnn
from bench import recall_at_k, synthetic
from ggnn_amd import _lib

ggnn.set_log_level(-1)
n, d, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
rest = sys.argv[4:]
base_kind = "lowrank16"   # --kind lowrank24: a harder synthetic base (long searches)
if rest and rest[0] == "--kind":
    base_kind, rest = rest[1], rest[2:]
# hook combinations to compare: --combos "QUERY_EARLY=0;QUERY_EARLY=1,QUERY_GLOBAL_RING=0;..."
combos = [{"QUERY_EARLY": 0}, {"QUERY_EARLY": 1}]
if rest and rest[0] == "--combos":
    combos = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in c.split(",") if kv)
              for c in rest[1].split(";")]
    rest = rest[2:]
points = [tuple(float(x) for x in p.split(":")) for p in rest] or [(0.85, 175)]
dev = torch.device("cuda", 0)
base = synthetic(base_kind, n, d, 1234, dev)
qs = {"tune": synthetic(base_kind, 10_000, d, 4321, dev),
      "held": synthetic(base_kind, 10_000, d, 8642, dev)}
big = synthetic(base_kind, 100_000, d, 9876, dev)
if kind == "u8":
    base, big = base.to(torch.uint8), big.to(torch.uint8)
    qs = {k: v.to(torch.uint8) for k, v in qs.items()}
lib = os.environ.get("GGNN_AMD_LIB", "default").split("/")[-1]
if n <= 2_000_000:
    # merge kernel order A/B (hook MERGE_EARLY): build time of fresh handles, twice each
    for early in (0, 1, 0, 1):
        e2 = ggnn.GGNN()
        e2.set_base_reference(base)
        with _lib.hooks(MERGE_EARLY=early):
            e2.build(24, 0.5, 2)
        print(f"{lib}: build with MERGE_EARLY={early}: {e2.last_timing_ms()['build_ms'] / 1e3:.3f} s",
              flush=True)
        del e2
eng = ggnn.GGNN()
eng.set_base_reference(base)
eng.set_return_results_on_gpu(True)
eng.build(24, 0.5, 2)
gts = {k: eng.bf_query(q, 10)[0] for k, q in qs.items()}
print(f"{lib}: {n} x {d} {kind}, build {eng.last_timing_ms()['build_ms'] / 1e3:.2f} s", flush=True)


def timed(x, tau, it, reps):
    for _ in range(2):
        eng.query(x, 10, tau, it)
    ms = []
    for _ in range(reps):
        out = eng.query(x, 10, tau, it)
        ms.append(eng.last_timing_ms()["query_ms"])
    return sum(ms) / len(ms), out


rows = []
for tau, it in points:
    it = int(it)
    row = {"tau": tau, "it": it}
    res = []
    for ci, combo in enumerate(combos):
        with _lib.hooks(**combo):
            eng.set_collect_counters(True)
            ids, dists = eng.query(qs["tune"], 10, tau, it)
            cnt, rr = eng.last_query_counters(), eng.last_query_rows_read()
            eng.set_collect_counters(False)
            ms10, _ = timed(qs["tune"], tau, it, 10)
            ms100, _ = timed(big, tau, it, 3)
            res.append((ids.clone(), dists.clone(), cnt, rr))
            tag = ",".join(f"{k}={v}" for k, v in combo.items()) or "default"
            row[tag] = {"ms10k": round(ms10, 4), "ms100k": round(ms100, 3)}
    row["identical"] = all(bool(torch.equal(r[0], res[0][0]) and torch.equal(r[1], res[0][1])
                                and r[2] == res[0][2] and r[3] == res[0][3]) for r in res)
    row["counters"] = res[0][2]
    row["rows"] = res[0][3]
    row["recall"] = {k: round(recall_at_k(eng.query(q, 10, tau, it)[0], gts[k]), 4)
                     for k, q in qs.items()}
    rows.append(row)
    print(json.dumps(row), flush=True)
