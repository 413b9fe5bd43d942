"""N>1 path of bench.py on CPU: two gloo ranks, each owning an env shard, agree on the whole-job
number exactly the way bench.py computes it (max-over-ranks time, sum of shard sizes), and the
shard-invariant reset noise keys give identical draws regardless of world size."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wall = 0.010 * (rank + 1)                       # rank 1 is the slow one
    value, wall_max = bench.aggregate_throughput(dist, torch.device("cpu"), wall, envs_per_rank=1000, steps=7)
    out[rank] = (value, wall_max)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_aggregation_gloo():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert len(out) == 2
    for r in range(world):
        value, wall_max = out[r]
        assert abs(wall_max - 0.020) < 1e-12                       # max over ranks
        assert abs(value - 2 * 1000 * 7 / 0.020) < 1e-6            # whole-job env-steps/s


def test_shard_invariant_env_ids():
    """bench.py gives rank r the global env ids [r*n, (r+1)*n): the union over ranks equals the
    single-process id range, so the Philox reset keys (seed, global env id, step) are independent of
    the number of shards."""
    sys.path.insert(0, ROOT)
    import bench
    n = 64
    one = bench.shard_env_ids(0, 1, 4 * n)
    four = np.concatenate([bench.shard_env_ids(r, 4, n) for r in range(4)])
    assert np.array_equal(one, four)
