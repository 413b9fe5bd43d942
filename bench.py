#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched Quadrotor hot path on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           --master-port 29500 bench.py --gpus 8 --steps 200 --warmup 20

Workload (BASELINE.json configs[1]): Quadrotor `hovering_control`, 65 536 parallel
envs per GPU, default config.json physics, dt=0.01 (10 Euler sub-steps per env step), nt=1000, flat
map; synthetic actions U(0.1, 15.0) f32 already resident in HBM; fused auto-reset so finished
episodes restart inside the launch. A "step" = ONE call of env.step() = one launch of the HIP
kernel over the whole batch. Envs shard across GPUs with no collective (weak scaling: 65 536
envs per GPU); torch.distributed (RCCL) is used only for the timing barrier and the max-over-ranks.

Prints ONE JSON line on rank 0.
  value        whole-job env-steps/s = n_gpus * 65536 * steps / max-over-ranks wall time
  roofline     algorithmic bytes per launch (317 B/env-step) / average kernel-launch
               duration measured with HIP events on the launching stream, vs the 8 TB/s HBM peak
  cpu_baseline the CPU oracle (a C port of the reference algorithm) on the host cores, bounded sample
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ENVS_PER_GPU = 65536
BYTES_PER_ENV_STEP = 317          # state R+W 2x116 + action 16 + obs 64 + reward 4 + done 1
HBM_PEAK_GBS = 8000.0             # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_ACTION_BATCHES = 8


def usable_cpus():
    """Host cores this process may really use: the affinity mask, capped by a cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(seconds=12.0, n_envs_per_thread=256, max_threads=None):
    """Time the CPU oracle (oracle/quadrotor_oracle.c, a scalar C port of the reference algorithm) on
    the same workload. One thread per host core, each stepping its own block of envs; the timed loop
    runs inside C (ctypes releases the GIL), 100 env-steps per env per call."""
    from oracle import quadrotor as qo
    cores = usable_cpus()
    if max_threads:
        cores = min(cores, max_threads)
    c = qo.default_consts()
    rs = np.random.RandomState(0)
    blocks = []
    for t in range(cores):
        n = n_envs_per_thread
        u = rs.random_sample((n, 4, 3))
        vel = (2.0 * u[:, 1]) * ((u[:, 0] > 0.5) * 2 - 1.0)
        om = (5.0 * u[:, 3]) * ((u[:, 2] > 0.5) * 2 - 1.0)
        st = qo.make_states(np.zeros((n, 3), np.float32), vel, om, np.zeros((n, 4), np.float32),
                            np.tile(np.eye(3, dtype=np.float32).reshape(9), (n, 1)))
        init = qo.make_states(np.zeros((n, 3), np.float32), vel, om, np.zeros((n, 4), np.float32),
                              np.tile(np.eye(3, dtype=np.float32).reshape(9), (n, 1)))
        acts = rs.uniform(0.1, 15.0, (N_ACTION_BATCHES, n, 4)).astype(np.float32)
        blocks.append((st, init, np.zeros(n, np.int32), acts))
    counts = [0] * cores
    stop = time.perf_counter() + seconds

    def work(i):
        st, init, ct, acts = blocks[i]
        total = 0
        while time.perf_counter() < stop:
            total += qo.batch_run(c, st, init, ct, acts, 100)
        counts[i] = total

    t0 = time.perf_counter()
    threads = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    el = time.perf_counter() - t0
    total = sum(counts)
    return {"value": total / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d env-steps of the same hovering_control workload (%d envs per thread, finished episodes "
                      "restart), %.1f s wall, C oracle gcc -O2 scalar, one thread per core"
                      % (total, n_envs_per_thread, el)}


def cpu_baseline_maze3d(seconds=4.0, res=256):
    """Secondary CPU figure for C3: the C oracle's MetaMazeDiscrete3D step + 256x256 render
    (oracle/maze_oracle.c, the scalar restatement of the un-jitted reference; the real numba timing is not
    available offline), one env per host core, for a few seconds."""
    from oracle import maze as mo
    from metagym_amd.metamaze import MAZE_TASK_MANAGER, MazeTaskSampler
    cores = usable_cpus()
    tt = mo.TASK_TYPES["SURVIVAL"]
    tex = MAZE_TASK_MANAGER.grounds.astype(np.uint8)
    counts = [0] * cores
    stop = time.perf_counter() + seconds

    def work(i):
        task = mo.Task(**MazeTaskSampler(n=9, allow_loops=False, step_reward=-0.01, goal_reward=1.0, food_density=0.06,
                                         food_interval=20, seed=i)._asdict())
        st = mo.State(task)
        mo.reset(task, tt, st)
        view = mo.View(tex, MAZE_TASK_MANAGER.ceil, res, res)
        rs = np.random.RandomState(i)
        n = 0
        while time.perf_counter() < stop:
            r, d = mo.step_disc3d(task, tt, 200, st, int(rs.randint(4)))
            mo.observe_3d(task, tt, view, st, 0)
            if d:
                mo.reset(task, tt, st)
            n += 1
        counts[i] = n

    t0 = time.perf_counter()
    threads = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    el = time.perf_counter() - t0
    return {"value": sum(counts) / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d steps + %dx%d frames, %.1f s wall, C oracle gcc -O2 scalar, one env per core"
                      % (sum(counts), res, res, el)}


def cpu_baseline_walker(seconds=3.0):
    """Secondary CPU figure for C4: the numpy restatement of the walker engine (oracle/abd.py — the checker the
    GPU kernels are tested against, NOT PyBullet, which the reference calls and which is not in its tree), one
    env on one core."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import abd
    from walker_fixtures import load_models
    m = load_models()["humanoid"]
    env = abd.WalkerEnv(m, prm=abd.Params(friction=0.8 * float(m.geom_friction), self_friction=float(m.geom_friction) ** 2))
    rs = np.random.RandomState(0)
    env.reset(rs.uniform(-0.1, 0.1, len(m.joint_lo)))
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        _, _, done, _ = env.step(rs.uniform(-1, 1, len(m.joint_lo)))
        if done:
            env.reset(rs.uniform(-0.1, 0.1, len(m.joint_lo)))
        n += 1
    el = time.perf_counter() - t0
    return {"value": n / el, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": "%d env steps of one humanoid, %.1f s wall, numpy restatement of this engine (oracle/abd.py)" % (n, el)}


def _time_steps(step_fn, steps, warmup):
    for i in range(warmup):
        step_fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        step_fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / steps


def secondary_workloads(dev):
    """The other BASELINE configs on this GPU, each a few dozen launches (reported next to the headline,
    never folded into `value`): C3 MetaMazeDiscrete3D 9x9 at the registered 256x256 resolution with
    16 384 envs, C1-scaled MetaMaze2D 15x15, C4 MetaLocomotion humanoid with 8 192 envs."""
    import metagym_amd
    from metagym_amd.metamaze import MazeTaskSampler
    out = {}
    try:
        n, res = 16384, 256
        tasks = [MazeTaskSampler(n=9, allow_loops=False, step_reward=-0.01, goal_reward=1.0, food_density=0.06,
                                 food_interval=20, seed=s) for s in range(64)]
        env = metagym_amd.make("meta-maze-discrete-3D-v0", num_envs=n, device=dev, max_steps=200,
                               resolution=(res, res), task_type="SURVIVAL", auto_reset=True)
        env.set_task(tasks)
        env.reset()
        acts = [torch.randint(0, 4, (n,), device=dev, dtype=torch.int32) for _ in range(4)]
        s = _time_steps(lambda i: env.step(acts[i % 4]), 12, 3)
        byt = (12 * res * res + 64) * n
        out["C3_maze3d_discrete_9x9_256x256_16384envs"] = {
            "env_steps_per_s": n / s, "ms_per_launch": s * 1e3,
            "roofline": {"bound": "hbm", "achieved": byt / s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": byt / s / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_env_step": 12 * res * res + 64}}
        del env, acts
        torch.cuda.empty_cache()
        n2 = 1 << 20
        tasks15 = [MazeTaskSampler(n=15, allow_loops=True, crowd_ratio=0.35, step_reward=-0.01, goal_reward=1.0,
                                   seed=s) for s in range(64)]
        env = metagym_amd.make("meta-maze-2D-v0", num_envs=n2, device=dev, max_steps=200, view_grid=1,
                               task_type="ESCAPE", auto_reset=True)
        env.set_task(tasks15)
        env.reset()
        acts = [torch.randint(0, 4, (n2,), device=dev, dtype=torch.int32) for _ in range(4)]
        s = _time_steps(lambda i: env.step(acts[i % 4]), 30, 5)
        out["C1_maze2d_15x15_escape_1048576envs"] = {"env_steps_per_s": n2 / s, "ms_per_launch": s * 1e3}
        del env, acts
        torch.cuda.empty_cache()
        # C1 as BASELINE.json states it: ONE 15x15 env. Pure launch latency — eager, and 100 steps
        # captured as one hipGraph (the C ABI only enqueues kernels, so torch.cuda.graph can capture it).
        env = metagym_amd.make("meta-maze-2D-v0", num_envs=1, device=dev, max_steps=10 ** 9, view_grid=1,
                               task_type="ESCAPE")
        env.set_task(tasks15[0])
        env.reset()
        a1 = torch.randint(0, 4, (100, 1), device=dev, dtype=torch.int32)
        s_eager = _time_steps(lambda i: env.step(a1[i % 100]), 200, 20)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            env.step(a1[0])
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for k in range(100):
                env.step(a1[k])
        s_graph = _time_steps(lambda i: graph.replay(), 20, 3) / 100
        out["C1_maze2d_15x15_escape_1env"] = {"us_per_step_eager": s_eager * 1e6,
                                              "us_per_step_hipgraph_100": s_graph * 1e6,
                                              "env_steps_per_s_hipgraph": 1.0 / s_graph}
        del env, graph
        torch.cuda.empty_cache()
    except Exception as e:  # secondary numbers must never break the headline line
        out["maze_error"] = repr(e)
    try:
        # C5 (mixed, 2^20 envs over 8 GPUs): one GPU's share = 65 536 quadrotors + 65 536 MetaMaze3D envs
        # (64x64 frames so the batch stays resident). The two families are independent,
        # so they are co-scheduled on two HIP streams: the VALU-bound quadrotor kernel and the
        # store-heavy raycaster overlap instead of queueing behind each other.
        nq = nm = 65536
        tasks = [MazeTaskSampler(n=9, allow_loops=False, step_reward=-0.01, goal_reward=1.0, food_density=0.06,
                                 food_interval=20, seed=s) for s in range(64)]
        quad = metagym_amd.make("quadrotor-v0", num_envs=nq, device=dev, task="hovering_control", auto_reset=True)
        maze = metagym_amd.make("meta-maze-discrete-3D-v0", num_envs=nm, device=dev, max_steps=200,
                                resolution=(64, 64), task_type="SURVIVAL", auto_reset=True)
        maze.set_task(tasks)
        quad.reset(seed=0)
        maze.reset()
        qa = [torch.rand(nq, 4, device=dev) * 14.9 + 0.1 for _ in range(4)]
        ma = [torch.randint(0, 4, (nm,), device=dev, dtype=torch.int32) for _ in range(4)]
        s_q = _time_steps(lambda i: quad.step(qa[i % 4]), 40, 5)
        s_m = _time_steps(lambda i: maze.step(ma[i % 4]), 40, 5)
        st_q, st_m = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

        def both(i):
            with torch.cuda.stream(st_q):
                quad.step(qa[i % 4])
            with torch.cuda.stream(st_m):
                maze.step(ma[i % 4])
        torch.cuda.synchronize()
        for i in range(5):
            both(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(40):
            both(i)
        torch.cuda.synchronize()
        s_b = (time.perf_counter() - t0) / 40
        out["C5_mixed_share_65536quad_plus_65536maze3d_64x64"] = {
            "env_steps_per_s_two_streams": (nq + nm) / s_b, "ms_per_mixed_step_two_streams": s_b * 1e3,
            "ms_quadrotor_alone": s_q * 1e3, "ms_maze3d_alone": s_m * 1e3,
            "overlap_gain": (s_q + s_m) / s_b}
        del quad, maze, qa, ma
        torch.cuda.empty_cache()
    except Exception as e:
        out["mixed_error"] = repr(e)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from walker_fixtures import load_models
        from metagym_amd.metalocomotion import MetaHumanoidEnv
        M = load_models()
        n = 8192
        env = MetaHumanoidEnv(num_envs=n, device=dev)
        env.set_task([M[k] for k in ("humanoid", "humanoid_tra_000", "humanoid_tra_137", "humanoid_ood_003")])
        env.reset(seed=0)
        acts = [torch.rand(n, env.n_joints, device=dev) * 2 - 1 for _ in range(4)]
        s = _time_steps(lambda i: env.step(acts[i % 4]), 12, 3)
        out["C4_humanoid_8192envs"] = {"env_steps_per_s": n / s, "ms_per_launch": s * 1e3,
                                       "note": "physics parity unpinned (PyBullet is not in the reference tree)"}
    except Exception as e:
        out["walker_error"] = repr(e)
    return out


def aggregate_throughput(dist, dev, wall, envs_per_rank, steps):
    """Whole-job env-steps/s: every rank stepped `envs_per_rank` envs `steps` times; the job took as
    long as its slowest rank. `dist` is torch.distributed (initialised) or None for one process."""
    wall_max, world = wall, 1
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall_max = float(t.item())
        world = dist.get_world_size()
    return world * envs_per_rank * steps / wall_max, wall_max


def shard_env_ids(rank, world, envs_per_rank):
    """Global env ids owned by `rank` (contiguous shards)."""
    return np.arange(rank * envs_per_rank, (rank + 1) * envs_per_rank, dtype=np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C1/C3/C4 side measurements")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":   # the env var lets a 1-GPU box test the RCCL path
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import metagym_amd
    n = args.envs_per_gpu
    env = metagym_amd.make("quadrotor-v0", num_envs=n, device=dev, task="hovering_control",
                           auto_reset=True, seed=1000, env_id_base=int(shard_env_ids(rank, world, n)[0]))
    env.reset(seed=1000 + rank)
    g = torch.Generator(device=dev)
    g.manual_seed(2000 + rank)
    actions = torch.rand(N_ACTION_BATCHES, n, 4, device=dev, generator=g) * 14.9 + 0.1

    def barrier():
        if dist is not None:
            dist.barrier()

    for i in range(args.warmup):
        env.step(actions[i % N_ACTION_BATCHES])
    barrier()
    torch.cuda.synchronize(dev)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                      # torch's current stream == the stream the kernels are launched on
    for i in range(args.steps):
        env.step(actions[i % N_ACTION_BATCHES])
    ev1.record()
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    barrier()
    wall = t1 - t0
    dev_ms = ev0.elapsed_time(ev1)

    value, wall_max = aggregate_throughput(dist, dev, wall, n, args.steps)

    done_frac = float(env._done.float().mean().item())
    failed_any = int(env._failed.max().item())
    if rank == 0:
        launch_s = dev_ms * 1e-3 / args.steps
        achieved = BYTES_PER_ENV_STEP * n / launch_s / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "quadrotor_pmc.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # The kernel is VALU-issue-bound, not HBM-bound: report how close the launch is to
        # the SIMDs' issue limit too. VALU instructions per wave come from the committed PMC pass
        # (profiles/<round>/pmc_summary.json, SQ_INSTS_VALU / SQ_WAVES); one wave-instruction occupies a
        # SIMD for >= 4 cycles, 1024 SIMDs at 2.4 GHz.
        valu = None
        try:
            rounds = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if d.startswith("r"))
            q = json.load(open(os.path.join(ROOT, "profiles", rounds[-1], "pmc_summary.json")))["quadrotor_step_kernel"]
            ipw = float(q["valu_insts_per_wave"])
            waves = (n + 63) // 64
            peak_issue = 1024 * 2.4e9 / 4.0
            valu = {"valu_insts_per_wave": ipw, "waves_per_launch": waves,
                    "achieved_wave_insts_per_s": ipw * waves / launch_s, "peak_wave_insts_per_s": peak_issue,
                    "frac": ipw * waves / launch_s / peak_issue, "source": "profiles/%s/pmc_summary.json" % rounds[-1]}
        except Exception:
            valu = None
        out = {
            "metric": "env-steps/sec (whole node) at 2^16 parallel envs; 1/2/4/8-GPU scaling",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32/f64 mixed (reference choreography)",
            "data": "synthetic",
            "config": {"workload": "Quadrotor hovering_control, %d envs/GPU, dt=0.01 (10 Euler sub-steps), "
                                   "nt=1000, flat map, actions U(0.1,15) f32, fused auto-reset" % n,
                       "launch": "one mg_quadrotor_step_autoreset launch per env.step()",
                       "envs_per_gpu": n, "sharding": "env-sharded, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_ceiling": achieved / 6290.0,
                         "traffic": traffic,
                         "kernel": "quadrotor_step_kernel", "avg_launch_us": launch_s * 1e6,
                         "algorithmic_bytes_per_launch": BYTES_PER_ENV_STEP * n,
                         "valu_issue": valu},
            "sanity": {"done_frac_last_step": done_frac, "failed_max": failed_any,
                       "host_wall_ms_per_step": wall / args.steps * 1e3},
        }
        if world == 1 and not args.no_secondary:
            del env
            torch.cuda.empty_cache()
            out["secondary"] = secondary_workloads(dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            if "secondary" in out and "C3_maze3d_discrete_9x9_256x256_16384envs" in out["secondary"]:
                try:
                    out["secondary"]["C3_maze3d_discrete_9x9_256x256_16384envs"]["cpu_baseline"] = cpu_baseline_maze3d()
                except Exception as e:
                    out["secondary"]["C3_maze3d_discrete_9x9_256x256_16384envs"]["cpu_baseline_error"] = repr(e)
            if "secondary" in out and "C4_humanoid_8192envs" in out["secondary"]:
                try:
                    out["secondary"]["C4_humanoid_8192envs"]["cpu_baseline"] = cpu_baseline_walker()
                except Exception as e:
                    out["secondary"]["C4_humanoid_8192envs"]["cpu_baseline_error"] = repr(e)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
