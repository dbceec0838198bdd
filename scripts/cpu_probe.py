"""host CPU facts behind bench.py's cpu_baseline: cgroup quota, affinity, single-thread rate of the
brute-force port and its scaling with threads"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except OSError:
        pass
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread|MHz' | head -8")
orc.set_fast_distance(True)
rng = np.random.default_rng(0)
base = rng.integers(0, 256, (200_000, 128)).astype(np.float32)
q = rng.integers(0, 256, (2048, 128)).astype(np.float32)
for th in (1, 8, 32, 64, 128, 256):
    n = min(2048, 32 * th)
    t = time.perf_counter(); orc.bf_query(base, q[:n], 10, threads=th); dt = time.perf_counter() - t
    print(f"threads {th:4d}: {n} queries x 200k rows in {dt:.3f} s -> {3.0*n*200000*128/dt/1e9:8.1f} GFLOP/s, {3.0*n*200000*128/dt/1e9/th:6.2f} per thread")
