"""8 resident shards x 1M on one GPU: query time with the shard launches overlapped / one at a time
(GGNN_TEST_HOOKS=1 GGNN_SHARD_OVERLAP=0: hooks are read from the environment only with the master switch)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ggnn_amd as ggnn
import bench
ggnn.set_log_level(-1)
class A: pass
a = A(); a.dataset = "lowrank16"; a.n_base = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
a.dim = 128; a.n_query = 10_000; a.k = 10; a.k_build = 24; a.tau_build = 0.5; a.refine = 2
a.tau_query = 0.9; a.max_iters = 175
print(bench.scaling_reference(a, torch.device("cuda", 0), ggnn, 10))
