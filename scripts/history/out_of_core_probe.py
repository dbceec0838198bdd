"""Out-of-core shards (hook RESIDENT_SHARDS) vs all-resident on 8 x 1M x 128 f32, one GPU:
query time of a 10k batch with the rows on the host (every swap re-uploads base shard + graph)
and with the rows resident on the GPU (only the graphs travel).
    python scripts/out_of_core_probe.py"""
import os, sys, json, time
sys.path.insert(0, os.getcwd())
import torch
import ggnn_amd as ggnn
from ggnn_amd import _lib
from bench import synthetic
ggnn.set_log_level(-1)
dev = torch.device("cuda", 0)
n_shard, shards = 1_000_000, 8
base = synthetic("lowrank16", n_shard * shards, 128, 1234, dev)
q = synthetic("lowrank16", 10_000, 128, 4321, dev)
host_base = base.cpu().pin_memory()
out = {}
for name, b, ref in (("rows on the GPU", base, True), ("rows on the (pinned) host", host_base, False)):
    for slots in (0, 2, 4):
        eng = ggnn.GGNN()
        if ref:
            eng.set_base_reference(b)
        else:
            eng.set_base(b)
        eng.set_shard_size(n_shard)
        t0 = time.perf_counter()
        with _lib.hooks(RESIDENT_SHARDS=slots):
            eng.build(24, 0.5, 2)
        build_s = time.perf_counter() - t0
        for _ in range(2):
            eng.query(q, 10, 0.9, 175)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ids, d = eng.query(q, 10, 0.9, 175)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        out[f"{name}, {'all resident' if not slots else str(slots) + ' GPU slots'}"] = {
            "ms_per_10k_batch": round(ms, 2), "build_wall_s": round(build_s, 2), "checksum": int(ids.sum())}
        del eng
        torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
