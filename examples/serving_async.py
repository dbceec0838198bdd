"""Batches in flight (serving extension, not part of the reference's surface):

    python examples/serving_async.py

`query()` blocks like the reference's (gpu_instance.cu:687-712).  A server that always has another
batch waiting can enqueue it with `query_async(..., slot=i)` before the previous one has finished:
batches on different slots run on different streams, so the under-occupied tail of one launch
overlaps with the head of the next (6.2 -> 8.8 M queries/s for 10k-query batches on one MI355X,
identical results)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ggnn_amd as ggnn  # noqa: E402

dev = torch.device("cuda", 0)
base = torch.randint(0, 256, (1_000_000, 128), device=dev).float()
batches = [torch.randint(0, 256, (10_000, 128), device=dev).float() for _ in range(8)]

g = ggnn.GGNN()
g.set_base(base)
g.set_return_results_on_gpu(True)
g.build(24, 0.5)

for q in batches[:2]:                       # warm-up
    g.query(q, 10, 0.6, 200)
torch.cuda.synchronize()
t = time.perf_counter()
blocking = [g.query(q, 10, 0.6, 200) for q in batches]
t_block = time.perf_counter() - t

t = time.perf_counter()
tickets = [g.query_async(q, 10, 0.6, 200, slot=i) for i, q in enumerate(batches)]
g.synchronize()                             # results are valid from here on
t_async = time.perf_counter() - t

assert all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(blocking, tickets))
n = sum(q.shape[0] for q in batches)
print(f"blocking: {n / t_block / 1e6:.2f} M queries/s   in flight: {n / t_async / 1e6:.2f} M queries/s")
