"""N>1 path of bench.py on CPU: two gloo ranks build their shards exactly the way bench.py does
(`bench.shard_plan`: global env ids, the `env_id_base` of the fused auto-reset, reset / action seeds, the maze
task of every env), STEP their shard, and agree on the whole-job number the way bench.py computes it
(max-over-ranks time, sum of shard sizes — over gloo, the backend bench.py itself uses: the harness has no
RCCL). The stepping runs on the CPU oracle's restatement of the fused-auto-reset launch (the HIP library needs
a GPU), which consumes the same plan fields the GPU env does: the union of the two ranks' trajectories must be
the single-rank job, bit for bit."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

N_PER_RANK, NT, T_STEPS, JOB_SEED = 48, 5, 13, 1000


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _initial_noise(env_ids):
    """Initial reset noise as a function of the GLOBAL env id (so that it does not depend on the sharding):
    draw e of RandomState(JOB_SEED) in the reference's order (sign_v, mag_v, sign_w, mag_w), quadrotorsim.py:241-254."""
    u = np.random.RandomState(JOB_SEED).random_sample((int(env_ids.max()) + 1, 4, 3))[env_ids]
    vel = (2.0 * u[:, 1]) * ((u[:, 0] > 0.5) * 2 - 1.0)
    om = (5.0 * u[:, 3]) * ((u[:, 2] > 0.5) * 2 - 1.0)
    return vel, om


def _actions(env_ids):
    a = np.random.RandomState(JOB_SEED + 1).uniform(0.1, 15.0, (T_STEPS, 4 * N_PER_RANK, 4)).astype(np.float32)
    return a[:, env_ids]


def _run_shard(plan):
    """Step one shard for T_STEPS with fused auto-reset; returns per-step obs / reward / done and the final counters."""
    from oracle import quadrotor as qo
    ids = plan["env_ids"]
    n = len(ids)
    vel, om = _initial_noise(ids)
    st = qo.make_states(np.zeros((n, 3), np.float32), vel, om, np.zeros((n, 4), np.float32),
                        np.tile(np.eye(3, dtype=np.float32).reshape(9), (n, 1)))
    ct, ep = np.zeros(n, np.int32), np.zeros(n, np.uint32)
    c = qo.default_consts(nt=NT)
    ar = qo.default_autoreset(seed=plan["job_seed"], env_id_base=plan["env_id_base"])
    acts = _actions(ids)
    rec = []
    for t in range(T_STEPS):
        obs, rew, done, failed = qo.batch_env_step_autoreset(c, ar, st, ct, ep, acts[t])
        rec.append((obs.copy(), rew.copy(), done.copy()))
    s = qo.states_to_arrays(st)
    return dict(obs=np.stack([r[0] for r in rec]), rew=np.stack([r[1] for r in rec]), done=np.stack([r[2] for r in rec]),
                vel=s["vel"], omega=s["omega"], episode=ep)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = bench.shard_plan(rank, world, N_PER_RANK, workload="mixed", job_seed=JOB_SEED)
    res = _run_shard(plan)
    wall = 0.010 * (rank + 1)                       # rank 1 is the slow one
    value, wall_max = bench.aggregate_throughput(dist, wall, envs_per_rank=2 * N_PER_RANK, steps=T_STEPS)
    out[rank] = dict(value=value, wall_max=wall_max, res=res, env_ids=plan["env_ids"], base=plan["env_id_base"],
                     maze_task_ids=plan["maze_task_ids"], reset_seed=plan["reset_seed"], action_seed=plan["action_seed"])
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_build_and_step_their_shards_gloo():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert len(out) == 2
    for r in range(world):
        assert abs(out[r]["wall_max"] - 0.020) < 1e-12                                    # max over ranks
        assert abs(out[r]["value"] - 2 * (2 * N_PER_RANK) * T_STEPS / 0.020) < 1e-6       # whole-job env-steps/s
    # the plans tile the job: contiguous global ids, bases, distinct host seeds, maze tasks by global id
    ids = np.concatenate([out[r]["env_ids"] for r in range(world)])
    assert np.array_equal(ids, np.arange(world * N_PER_RANK))
    assert [out[r]["base"] for r in range(world)] == [0, N_PER_RANK]
    assert len({out[r]["reset_seed"] for r in range(world)}) == world
    assert len({out[r]["action_seed"] for r in range(world)}) == world
    assert np.array_equal(np.concatenate([out[r]["maze_task_ids"] for r in range(world)]), ids % 64)
    # the union of the two ranks' trajectories is the single-rank job (Philox keyed by global env id + episode)
    sys.path.insert(0, ROOT)
    import bench
    single = _run_shard(bench.shard_plan(0, 1, world * N_PER_RANK, workload="mixed", job_seed=JOB_SEED))
    for k, axis in (("obs", 1), ("rew", 1), ("done", 1), ("vel", 0), ("omega", 0), ("episode", 0)):
        joined = np.concatenate([out[r]["res"][k] for r in range(world)], axis=axis)
        assert np.array_equal(joined, single[k]), k
    assert int(single["episode"].min()) == T_STEPS // NT                                   # episodes did restart


def _solo_worker(rank, world, port, out):
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t0 = time.perf_counter()
    solo = bench.solo_wall_of(dist, rank, lambda: time.sleep(0.05))          # rank 0 "steps" alone for 50 ms
    waited = time.perf_counter() - t0
    wall = 0.010 * (rank + 1)                                                # the all-rank region: rank 1 is the slow one
    value, _ = bench.aggregate_throughput(dist, wall, envs_per_rank=N_PER_RANK, steps=T_STEPS)
    walls = bench.gather_walls(dist, wall)
    fields = bench.scaling_fields(value, world, N_PER_RANK, T_STEPS, walls, solo_wall=solo, single_gpu_value=1000.0) if rank == 0 else None
    out[rank] = dict(solo=solo, waited=waited, fields=fields, value=value)
    dist.barrier()
    dist.destroy_process_group()


def test_self_baseline_and_per_rank_values_at_world_size_two():
    """`bench.py --gpus N --self-baseline`: rank 0 times the K steps alone while the other ranks wait at a
    barrier, and the line carries per-rank values and efficiency = value(N) / (N x solo) — the protocol over gloo, world size 2."""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_solo_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[1]["solo"] is None and 0.05 <= out[0]["solo"] < 0.5
    assert out[1]["waited"] >= 0.045                                         # rank 1 did idle through rank 0's solo run
    f = out[0]["fields"]
    assert np.allclose(f["rank_wall_ms"], [10.0, 20.0])
    assert np.allclose(f["rank_value"], [N_PER_RANK * T_STEPS / 0.010, N_PER_RANK * T_STEPS / 0.020])
    solo_value = N_PER_RANK * T_STEPS / out[0]["solo"]
    assert abs(f["self_baseline"]["value"] - solo_value) < 1e-6 * solo_value
    assert abs(f["efficiency"] - out[0]["value"] / (2 * solo_value)) < 1e-12
    assert abs(f["efficiency_vs_single_gpu_value"] - out[0]["value"] / 2000.0) < 1e-12
    # without a solo run the prior-N=1 route keeps the plain name; and the flag parses on the plan-only path
    sys.path.insert(0, ROOT)
    import bench
    g = bench.scaling_fields(100.0, 2, 8, 5, [0.1, 0.2], single_gpu_value=60.0)
    assert abs(g["efficiency"] - 100.0 / 120.0) < 1e-12 and "self_baseline" not in g
    bench.main(["--gpus", "1", "--plan-only", "--self-baseline", "--envs-per-gpu", "64"])


def test_shard_invariant_env_ids():
    """bench.py gives rank r the global env ids [r*n, (r+1)*n): the union over ranks equals the
    single-process id range, so the Philox reset keys (seed; global env id, episode) are independent of
    the number of shards."""
    sys.path.insert(0, ROOT)
    import bench
    n = 64
    one = bench.shard_env_ids(0, 1, 4 * n)
    four = np.concatenate([bench.shard_env_ids(r, 4, n) for r in range(4)])
    assert np.array_equal(one, four)
    plans = [bench.shard_plan(r, 8, 65536, workload="mixed") for r in range(8)]
    assert sum(len(p["env_ids"]) + len(p["maze_env_ids"]) for p in plans) == 1 << 20        # C5: 2^20 envs on 8 GPUs
    assert [p["env_id_base"] for p in plans] == [r * 65536 for r in range(8)]


def test_bare_gpus_flag_self_spawns_ranks_and_reaches_the_plan_stage(capfd):
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run around it (the form the driver uses for N = 1): bench.main
    spawns the two ranks itself (LOCAL_RANK / RANK / WORLD_SIZE, a free port), they rendezvous over gloo, build their
    shard plans and rank 0 prints one JSON line. `--plan-only` stops before any GPU work, so this runs here."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    saved = {k: os.environ.pop(k, None) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    try:
        bench.main(["--gpus", "2", "--plan-only", "--envs-per-gpu", "4096"])
    finally:
        for k, v in saved.items():
            if v is not None:
                os.environ[k] = v
    lines = [l for l in capfd.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["spawned"] and rec["union_is_single_job"]
    assert rec["shards"] == [[0, 4096, 0], [4096, 8192, 4096]]
