import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ggnn_amd as ggnn
from bench import synthetic
ggnn.set_log_level(-1)
dev = torch.device("cuda", 0)
base = synthetic("lowrank16", 1_000_000, 128, 1234, dev)
q_dev = synthetic("lowrank16", 10_000, 128, 4321, dev)
q_host = q_dev.cpu().contiguous(); q_pinned = q_host.pin_memory()
eng = ggnn.GGNN(); eng.set_base_reference(base); eng.build(24, 0.5, 2)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - s) / n * 1e3
eng.set_return_results_on_gpu(True)
print("device in / device out :", round(t(lambda: eng.query(q_dev, 10, 0.9, 200)), 3), "ms")
eng.set_return_results_on_gpu(False)
print("device in / host out   :", round(t(lambda: eng.query(q_dev, 10, 0.9, 200)), 3), "ms")
print("host in / host out     :", round(t(lambda: eng.query(q_host, 10, 0.9, 200)), 3), "ms")
print("pinned host in / host out:", round(t(lambda: eng.query(q_pinned, 10, 0.9, 200)), 3), "ms")
