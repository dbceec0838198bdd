import os, sys, json
sys.path.insert(0, os.getcwd())
import torch
import ggnn_amd as ggnn
from ggnn_amd import _lib
from bench import synthetic, recall_at_k
ggnn.set_log_level(-1)
n, d, kind, dt = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
tau, it = sys.argv[5].split(":"); tau, it = float(tau), int(it)
dev = torch.device("cuda", 0)
base = synthetic(kind, n, d, 1234, dev); q = synthetic(kind, 10000, d, 4321, dev); big = synthetic(kind, 100000, d, 9876, dev)
if dt == "u8":
    base, q, big = base.to(torch.uint8), q.to(torch.uint8), big.to(torch.uint8)
for c in (1, 0, 1, 0):
    with _lib.hooks(MERGE_COUNTING=c):
        e = ggnn.GGNN(); e.set_base_reference(base); e.build(24, 0.5, 2)
        print(f"MERGE_COUNTING={c} build {e.last_timing_ms()['build_ms']/1e3:.4f} s", flush=True)
    del e
eng = ggnn.GGNN(); eng.set_base_reference(base); eng.set_return_results_on_gpu(True); eng.build(24, 0.5, 2)
res = {}
for rep in range(2):
  for c in (1, 0):
    with _lib.hooks(MERGE_COUNTING=c):
        for x, name, reps in ((q, "10k", 10), (big, "100k", 3)):
            for _ in range(2): eng.query(x, 10, tau, it)
            ms = []
            for _ in range(reps):
                out = eng.query(x, 10, tau, it); ms.append(eng.last_timing_ms()["query_ms"])
            res[(c, name)] = out
            print(f"MERGE_COUNTING={c} {name} {sum(ms)/len(ms):.4f} ms", flush=True)
print("identical", all(torch.equal(res[(1, k)][0], res[(0, k)][0]) and torch.equal(res[(1, k)][1], res[(0, k)][1]) for k in ("10k", "100k")))
