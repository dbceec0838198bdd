import sys, os, json
sys.path.insert(0, os.getcwd())
import torch
import ggnn_amd as ggnn
from bench import synthetic, recall_at_k
ggnn.set_log_level(-1)
dev = torch.device("cuda", 0)
base = synthetic("lowrank16", 1_000_000, 128, 1234, dev)
q = synthetic("lowrank16", 10_000, 128, 4321, dev)
big = synthetic("lowrank16", 100_000, 128, 9876, dev)
eng = ggnn.GGNN(); eng.set_base_reference(base); eng.set_return_results_on_gpu(True)
eng.build(24, 0.5, 2)
def t(x, n=20):
    for _ in range(3): eng.query(x, 10, 0.9, 175)
    ms=[]
    for _ in range(n):
        ids,_=eng.query(x, 10, 0.9, 175); ms.append(eng.last_timing_ms()["query_ms"])
    return sum(ms)/len(ms), int(ids.sum())
a=t(q); b=t(big,5)
print(os.environ.get("GGNN_AMD_LIB","default").split("/")[-1], "10k: %.4f ms  100k: %.3f ms  checksum %d" % (a[0], b[0], a[1]), "build %.3f" % (eng.last_timing_ms()["build_ms"]/1e3))
