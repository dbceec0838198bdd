"""Run under torch.distributed.run with 2+ ranks (tests/test_gpu_distributed.py): the REAL engine
behind ShardedGGNN on every rank (all ranks share GPU 0 on the one-GPU test box, so the exchange
uses the gloo backend and stages the small candidate lists through the host).  Rank 0 writes the
merged results to argv[1]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    out, shard, spg = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    from conftest import make_int_data
    from ggnn_amd.distributed import ShardedGGNN
    N, D, K = shard * spg * world, 64, 10
    base = make_int_data(N, D, 871)
    q = make_int_data(200, D, 872)
    sh = ShardedGGNN()
    sh.set_base(torch.from_numpy(base).cuda())      # the whole base: each rank keeps its slice
    sh.set_shard_size(shard)
    sh.build(24, 0.5, 1)
    qd = torch.from_numpy(q).cuda()
    gt, gt_d = sh.bf_query(qd, K)
    ids, d = sh.query(qd, K, 0.8, 400)
    # batches in flight: tickets finished in order give the blocking results
    t0 = sh.query_async(qd, K, 0.8, 400, slot=0)
    t1 = sh.query_async(qd, K, 0.8, 400, slot=1)
    for t in (t0, t1):
        a_ids, a_d = sh.finish(t)
        assert torch.equal(a_ids, ids) and torch.equal(a_d, d)
    local_ids, local_d = sh.engine.query(qd, K, 0.8, 400)   # this rank's sorted [Nq, K*spg] rows
    gathered = [None] * world
    dist.all_gather_object(gathered, (local_ids.cpu().numpy(), local_d.cpu().numpy()))
    if rank == 0:
        np.savez(out, gt=gt.cpu().numpy(), gt_d=gt_d.cpu().numpy(), ids=ids.cpu().numpy(),
                 d=d.cpu().numpy(), parts_ids=np.stack([g[0] for g in gathered]),
                 parts_d=np.stack([g[1] for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
