#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched MetaGym hot path on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           --master-port 29500 bench.py --gpus 8 --steps 200 --warmup 20 [--workload mixed]

Headline workload (BASELINE.json configs[1]): Quadrotor `hovering_control`, 65 536
parallel envs per GPU, default config.json physics, dt=0.01 (10 Euler sub-steps per env step), nt=1000,
flat map; synthetic actions U(0.1, 15.0) f32 already resident in HBM; fused auto-reset so finished
episodes restart inside the launch. A "step" = ONE env.step() over the whole batch = ONE launch of the HIP
kernel. `--launch graph` captures the K timed env.step() calls into one hipGraph and times
its replay — the fused auto-reset draws its noise from per-env device counters, so no launch argument
changes between steps; `--launch eager` issues the K calls from Python. Same kernel, same work per step.
`--launch auto` (default) times BOTH, each in its own barrier-bracketed region of exactly K steps, and reports the
mode with the higher whole-job throughput (the other under sanity.other_launch_mode): a short graph (the driver's
K = 20) does not amortise its ~40 us launch, a long one beats the Python loop only when the host is slow.

`--workload mixed` (BASELINE.json configs[4]): every rank steps 65 536 quadrotors AND
65 536 MetaMazeDiscrete3D envs (9x9 mazes, 64x64 frames so 2^19 frames stay resident) on two HIP streams;
at N = 8 that is 2^20 envs. A "step" = one step of both families.

Envs shard across GPUs with no data-path collective (weak scaling). torch.distributed is initialised with
the **gloo** backend and used only for the timing barrier and the max-over-ranks reduction of one float:
there is no RCCL anywhere in the harness or the step path (north_star: "independent batches, no RCCL").

Prints ONE JSON line on rank 0.
  value        whole-job env-steps/s = envs stepped by all ranks * steps / max-over-ranks wall time
  roofline     algorithmic bytes per launch (317 B/env-step) / average kernel-launch
               duration measured with HIP events on the launching stream, vs the 8 TB/s HBM peak
  cpu_baseline the UNMODIFIED reference Quadrotor.step, one single-env worker process per host core (kind "reference":
               imported from oracle/_ref, the byte-compiled copy oracle/make_ref.py builds and ships); the C port of it
               (oracle/) beside it as cpu_port — and alone (kind "port") only if oracle/_ref is missing
"""
import argparse
import ctypes
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
COPY_CEILING_GBS = 6290.0         # same guide: what a device copy achieves
FP64_VALU_PEAK_TFLOPS = 78.6      # same guide: 256 CUs x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
N_ACTION_BATCHES = 8
METRIC = "env-steps/sec (whole node) at 2^16 parallel envs; 1/2/4/8-GPU scaling"


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


# ---------------------------------------------------------------------------------------- CPU baselines
def cpu_port(seconds=12.0, n_envs_per_thread=256, max_threads=None):
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


def reference_root():
    """Where the unmodified reference can be imported from: $METAGYM_REFERENCE, else oracle/_ref — the byte-compiled
    reference packages oracle/make_ref.py builds in the build container (git-ignored; it travels to the GPU box with the
    snapshot like the built .so files; bench.py never reads /root/reference at run time)."""
    env = os.environ.get("METAGYM_REFERENCE")
    if env:
        return env
    return os.path.join(ROOT, "oracle", "_ref")


def _reference_has(ref, pkg):
    d = os.path.join(ref, "metagym", pkg)
    return os.path.isfile(os.path.join(d, "__init__.py")) or os.path.isfile(os.path.join(d, "__init__.pyc"))


def _reference_pool(kind, ref, seconds, cores, *extra):
    """`cores` independent single-env worker PROCESSES of the unmodified reference (oracle/ref_workers.py, one fresh interpreter
    each: this process has torch and a live HIP runtime loaded, which a fork would copy and a multiprocessing spawn would
    re-import per worker), all started together; -> [(steps, seconds)] per worker. A worker that does not report within
    `seconds` + 120 s is killed and the whole leg fails loudly rather than hanging the bench."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "ref_workers.py"), kind]
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen(cmd + [str(i), ref, repr(float(seconds))] + list(extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, cwd=ROOT, env=env) for i in range(cores)]
    out = []
    try:
        for p in procs:
            so, se = p.communicate(timeout=seconds + 120.0)
            if p.returncode != 0:
                raise RuntimeError("reference worker failed: %s" % se.strip()[-400:])
            n, t = so.split()[-2:]
            out.append((int(n), float(t)))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return out


def cpu_reference(seconds=12.0):
    """north_star: "the reference CPU path timed on the same box's host cores (core count stated) in the same
    run" — P independent single-env `Quadrotor.step` worker processes, P = the usable cores (BASELINE.md §4,
    SURVEY.md §8d C2: "env.step in a multiprocessing.Pool(P) of independent single-env workers"), imported from
    reference_root() (oracle/_ref on the GPU box): returns (result, None) or (None, reason)."""
    ref = reference_root()
    if not _reference_has(ref, "quadrotor"):
        return None, ("no importable reference at %s (oracle/make_ref.py builds it in the build container; "
                      "__graft_entry__.build() calls it); the figure recorded in the build container is under "
                      "from_profiles.reference_cpu" % ref)
    cores = usable_cpus()
    try:
        res = _reference_pool("quadrotor", ref, seconds, cores)
    except Exception as e:
        return None, "reference import / run failed: %r" % (e,)
    steps = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    return {"value": steps / wall, "unit": "env-steps/s", "cores": cores, "kind": "reference",
            "per_core": steps / wall / cores,
            "source": ("oracle/_ref (byte-code of the unmodified reference, oracle/make_ref.py)"
                       if os.path.abspath(ref) == os.path.join(ROOT, "oracle", "_ref") else ref),
            "sample": "%d env-steps, %d single-env worker processes of the unmodified "
                      "metagym.quadrotor.env.Quadrotor.step (hovering_control, dt=0.01, nt=1000, U(0.1,15) actions, finished "
                      "episodes reset; numpy %s, gym stubbed), %.1f s each" % (steps, cores, np.__version__, wall)}, None


def cpu_reference_maze2d(seconds=2.0):
    """BASELINE.json configs[0]: "MetaMaze2D 15x15 discrete, 1 env, reference CPU step()" — ONE single-env worker process of the
    unmodified MetaMaze2D (maze_env.py:189-204) per task type, on one host core (the configuration is one env; more cores would
    be more envs)."""
    ref = reference_root()
    if not _reference_has(ref, "metamaze"):
        return None
    out = {"cores": 1, "kind": "reference", "unit": "us per env.step() (episode resets inside the clock)"}
    for tt in ("ESCAPE", "SURVIVAL"):
        try:
            (steps, wall), = _reference_pool("maze2d", ref, seconds, 1, tt)
            out[tt] = {"us_per_step": wall / steps * 1e6, "env_steps_per_s": steps / wall,
                       "sample": "%d steps of one unmodified MetaMaze2D (15x15, crowd_ratio 0.35, view_grid 1, max_steps 200, uniform "
                                 "random actions), %.1f s" % (steps, wall)}
        except Exception as e:
            out[tt] = {"error": repr(e)}
    return out


def cpu_reference_maze3d(seconds=5.0, res=256):
    """C3 "for the record": the unmodified reference MetaMazeDiscrete3D.step (+ its 256x256 frame) on every host core.
    numba is absent, so the ray caster runs un-jitted — this is NOT what a user of the reference with numba would see."""
    ref = reference_root()
    if not _reference_has(ref, "metamaze"):
        return None
    cores = usable_cpus()
    try:
        out = _reference_pool("maze3d", ref, seconds, cores, str(res))
    except Exception as e:
        return {"error": repr(e)}
    steps, wall = sum(r[0] for r in out), max(r[1] for r in out)
    return {"value": steps / wall, "unit": "env-steps/s", "cores": cores, "kind": "reference (numba absent: un-jitted Python)",
            "sample": "%d steps + %dx%d frames, %d single-env workers of the unmodified MetaMazeDiscrete3D, %.1f s each"
                      % (steps, res, res, cores, wall)}


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


def cpu_baseline_walker(seconds=10.0, envs_per_thread=32):
    """CPU figure for C4 ("own C++ restatement on P cores"): oracle/walker_oracle.c — this engine's env step
    in scalar C (the wave kernel's algorithm; pinned to the numpy restatement oracle/abd.py to 1e-12 per sub-step by
    tests/test_oracle_walker_c.py), NOT PyBullet, which the reference calls and which is not in its tree. One thread per usable
    core, each stepping its own humanoids (the 256 TRAIN variants round-robin, random actions, finished episodes restart); the
    timed loop runs inside C (ctypes releases the GIL)."""
    import ctypes as C
    from oracle import abd, walker_c
    from metagym_amd.metalocomotion import variants
    lib = walker_c.load()
    cores = usable_cpus()
    models = variants.models("humanoid", "TRAIN")
    power = abd.HUMANOID_MOTOR_POWER * 0.41
    packed = [walker_c.make_model(m, power) for m in models]
    cmodels = (walker_c.Model * len(packed))(*[p[0] for p in packed])
    prm = walker_c.humanoid_params(models[0], max_steps=1000)
    rs = np.random.RandomState(0)
    actions = np.ascontiguousarray(rs.uniform(-1, 1, (4096, 17)).astype(np.float32))
    blocks = []
    for t in range(cores):
        envs = (walker_c.Env * envs_per_thread)()
        ids = (C.c_int * envs_per_thread)(*[(t * envs_per_thread + i) % len(models) for i in range(envs_per_thread)])
        for i in range(envs_per_thread):
            noise = np.ascontiguousarray(rs.uniform(-0.1, 0.1, 17))
            lib.wo_env_reset(C.byref(cmodels[ids[i]]), C.byref(prm), C.byref(envs[i]), noise.ctypes.data_as(C.POINTER(C.c_double)), None)
        blocks.append((envs, ids))
    counts = [0] * cores
    stop = time.perf_counter() + seconds

    def work(i):
        envs, ids = blocks[i]
        total = 0
        while time.perf_counter() < stop:
            total += lib.wo_run(cmodels, ids, C.byref(prm), envs, envs_per_thread, 8, actions.ctypes.data_as(C.POINTER(C.c_float)), 4096)
        counts[i] = total

    t0 = time.perf_counter()
    threads = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    el = time.perf_counter() - t0
    return {"value": sum(counts) / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d env steps of humanoids over the 256 TRAIN variants (%d envs per thread, random actions, finished episodes "
                      "restart), %.1f s wall, oracle/walker_oracle.c gcc -O2 scalar, one thread per core; this engine's algorithm, not "
                      "PyBullet (absent from the reference tree)" % (sum(counts), envs_per_thread, el)}


def walker_flops(robot="humanoid", full=False):
    """Counted f64 operations per env step (oracle/count_walker_flops.py: the instrumented C restatement of the wave kernel's
    algorithm over a rollout of this workload), from the newest committed profiles/<round>/walker_flops.json that has the entry
    (`humanoid`, `ant`: BASELINE C4's airborne rollout; `humanoid_grounded`, `humanoid_lying`: the contact-rich ones).
    full=True also returns the whole record (constraint rows / contacts per sub-step)."""
    none = (None, None, None, None) if full else (None, None, None)
    for rnd in sorted((d for d in os.listdir(os.path.join(ROOT, "profiles")) if d.startswith("r") and d[1:].isdigit()), reverse=True):
        p = os.path.join(ROOT, "profiles", rnd, "walker_flops.json")
        if os.path.exists(p):
            rec = json.load(open(p)).get(robot)
            if rec is None:
                continue
            res = (float(rec["flop_per_env_step"]), "profiles/%s/walker_flops.json (%s)" % (rnd, rec["source"]), rec["per_env_step"])
            return res + (rec,) if full else res
    return none


# ---------------------------------------------------------------------------------------- GPU timing helpers
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


def capture_steps(dev, step_fn, steps):
    """`steps` calls of step_fn(i) as one hipGraph (torch.cuda.graph). The C ABI only enqueues kernels on the
    caller's stream, so it captures as it stands. Returns the graph; replaying it performs exactly those steps."""
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):          # lazy module load must not happen during capture
        step_fn(0)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(steps):
            step_fn(i)
    return graph


def maze_tasks(n_tasks=64):
    from metagym_amd.metamaze import MazeTaskSampler
    return [MazeTaskSampler(n=9, allow_loops=False, step_reward=-0.01, goal_reward=1.0, food_density=0.06,
                            food_interval=20, seed=s) for s in range(n_tasks)]


def latest_profile_round(holding="pmc_summary.json"):
    """The newest profiles/rNN/ that holds `holding`."""
    try:
        rounds = sorted((d for d in os.listdir(os.path.join(ROOT, "profiles")) if d.startswith("r") and d[1:].isdigit()), reverse=True)
        return next((r for r in rounds if os.path.exists(os.path.join(ROOT, "profiles", r, holding))), None)
    except OSError:
        return None


def from_profiles(n, launch_s):
    """Numbers that were NOT observed in this run: read from committed rocprofv3 / PMC summaries under profiles/
    (file and round named next to each). Kept apart from `roofline`, which holds only what this run measured."""
    out = {}
    rnd = latest_profile_round()
    if rnd is None:
        return out
    try:
        q = json.load(open(os.path.join(ROOT, "profiles", rnd, "pmc_summary.json")))["quadrotor_step_kernel"]
        src = "profiles/%s/pmc_summary.json" % rnd
        if "hbm_bytes_per_launch" in q:
            out["traffic"] = {"hbm_bytes_per_launch": q["hbm_bytes_per_launch"], "source": src, "round": rnd,
                              "note": "2 x FETCH_SIZE + WRITE_SIZE of a separate --pmc pass at the same batch size"}
        ipw = float(q["valu_insts_per_wave"])
        waves = (n + 63) // 64
        peak_issue = 1024 * 2.4e9 / 4.0           # one wave-instruction occupies a SIMD for >= 4 cycles
        out["valu_issue"] = {"valu_insts_per_wave": ipw, "waves_per_launch": waves,
                             "achieved_wave_insts_per_s": ipw * waves / launch_s,
                             "peak_wave_insts_per_s": peak_issue, "frac": ipw * waves / launch_s / peak_issue,
                             "source": src, "round": rnd,
                             "note": "instruction count from the PMC pass, launch duration from this run"}
    except Exception:
        pass
    try:
        p = os.path.join(ROOT, "profiles", rnd, "reference_cpu_build_container.json")
        if not os.path.exists(p):
            p = None
            for r in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
                cand = os.path.join(ROOT, "profiles", r, "reference_cpu_build_container.json")
                if os.path.exists(cand):
                    p = cand
                    break
        if p:
            out["reference_cpu"] = dict(json.load(open(p)), source=os.path.relpath(p, ROOT))
    except Exception:
        pass
    return out


def north_star_quadrotor(dev, sizes=(131072, 1048576), steps=60, warmup=10, valu_insts_per_wave=None):
    """north_star's own configuration — Quadrotor hovering_control at 2^20 parallel envs on 8 GPUs, i.e. 2^17 per GPU —
    and the whole 2^20 batch on ONE GPU, through the same QuadrotorShard (staggered clocks, steady-state pre-roll, fused
    auto-reset) as the headline. Both roofline fractions: HBM on the 317 B/env-step of SURVEY.md §8(d), and VALU issue
    from the per-wave instruction count of the committed PMC pass (a wave-instruction holds a SIMD for >= 4 cycles)."""
    out = {}
    ipw, src = valu_insts_per_wave, "this run's rocprofv3 --pmc SQ_INSTS_VALU / SQ_WAVES pass at the headline batch size (the count is per wave)"
    rnd = latest_profile_round()
    for r in ([] if ipw else ([rnd] if rnd else []) + ["r03", "r02"]):
        try:
            ipw = float(json.load(open(os.path.join(ROOT, "profiles", r, "pmc_summary.json")))["quadrotor_step_kernel"]["valu_insts_per_wave"])
            src = "profiles/%s/pmc_summary.json" % r
            break
        except Exception:
            continue
    for n in sizes:
        plan = shard_plan(0, 1, n, "quadrotor")
        q = QuadrotorShard(dev, plan, n)
        ep0 = q.episodes()
        s = _time_steps(q.step, steps, warmup)
        ep1 = q.episodes()
        gbs = BYTES_PER_ENV_STEP * n / s / 1e9
        e = {"env_steps_per_s": n / s, "us_per_launch": s * 1e6,
             "done_frac_per_step": (ep1 - ep0) / float(n * (steps + warmup)),
             "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                          "algorithmic_bytes_per_launch": BYTES_PER_ENV_STEP * n}}
        if ipw:
            waves = (n + 63) // 64
            peak_issue = 1024 * 2.4e9 / 4.0
            e["valu_issue"] = {"valu_insts_per_wave": ipw, "waves_per_launch": waves, "waves_per_simd": waves / 1024.0,
                               "achieved_wave_insts_per_s": ipw * waves / s, "peak_wave_insts_per_s": peak_issue,
                               "frac": ipw * waves / s / peak_issue, "source": src,
                               "note": "instruction count per wave from the PMC pass named in `source`, launch duration from this run"}
        out["north_star_quadrotor_hovering_%denvs_1gpu" % n] = e
        del q
        torch.cuda.empty_cache()
    return out


def secondary_workloads(dev, valu_insts_per_wave=None):
    """The other BASELINE configs on this GPU, each a few dozen launches (reported next to the headline,
    never folded into `value`): C3 MetaMazeDiscrete3D 9x9 at the registered 256x256 resolution with
    16 384 envs, C1 MetaMaze2D 15x15, C4 MetaLocomotion humanoid with 8 192 envs over the 256 TRAIN variants."""
    import metagym_amd
    from metagym_amd.metamaze import MazeTaskSampler
    out = {}
    try:
        out.update(north_star_quadrotor(dev, valu_insts_per_wave=valu_insts_per_wave))
    except Exception as e:
        out["north_star_error"] = repr(e)
    try:
        n, res = 16384, 256
        for cont, ident, key in ((False, "meta-maze-discrete-3D-v0", "C3_maze3d_discrete_9x9_256x256_16384envs"),
                                 (True, "meta-maze-continuous-3D-v0", "C3_maze3d_continuous_9x9_256x256_16384envs")):
            env = metagym_amd.make(ident, num_envs=n, device=dev, max_steps=200, resolution=(res, res),
                                   task_type="SURVIVAL", auto_reset=True)
            env.set_task(maze_tasks())
            env.reset()
            if cont:
                acts = [torch.rand(n, 2, device=dev) * 2 - 1 for _ in range(4)]
            else:
                acts = [torch.randint(0, 4, (n,), device=dev, dtype=torch.int32) for _ in range(4)]
            s = _time_steps(lambda i: env.step(acts[i % 4]), 12, 3)
            byt = (12 * res * res + 64) * n
            out[key] = {"env_steps_per_s": n / s, "ms_per_launch": s * 1e3,
                        "roofline": {"bound": "hbm", "achieved": byt / s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": byt / s / 1e9 / HBM_PEAK_GBS,
                                     "algorithmic_bytes_per_env_step": 12 * res * res + 64}}
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
        # captured as one hipGraph.
        env = metagym_amd.make("meta-maze-2D-v0", num_envs=1, device=dev, max_steps=10 ** 9, view_grid=1,
                               task_type="ESCAPE")
        env.set_task(tasks15[0])
        env.reset()
        a1 = torch.randint(0, 4, (100, 1), device=dev, dtype=torch.int32)
        s_eager = _time_steps(lambda i: env.step(a1[i % 100]), 200, 20)
        graph = capture_steps(dev, lambda i: env.step(a1[i]), 100)
        s_graph = _time_steps(lambda i: graph.replay(), 20, 3) / 100
        out["C1_maze2d_15x15_escape_1env"] = {"us_per_step_eager": s_eager * 1e6,
                                              "us_per_step_hipgraph_100": s_graph * 1e6,
                                              "env_steps_per_s_hipgraph": 1.0 / s_graph}
        del env, graph
        torch.cuda.empty_cache()
    except Exception as e:  # secondary numbers must never break the headline line
        out["maze_error"] = repr(e)
    try:
        # C5 (mixed, 2^20 envs over 8 GPUs): one GPU's share, same code path as `--workload mixed`
        n = ENVS_PER_GPU
        plan = shard_plan(0, 1, n, "mixed")
        quad, maze = QuadrotorShard(dev, plan, n), MazeShard(dev, plan, n)
        s_q = _time_steps(quad.step, 40, 5)
        s_m = _time_steps(maze.step, 40, 5)
        both = MixedStep(dev, quad, maze)
        for i in range(5):
            both(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(40):
            both(i)
        torch.cuda.synchronize()
        s_b = (time.perf_counter() - t0) / 40
        out["C5_mixed_share_65536quad_plus_65536maze3d_64x64"] = {
            "env_steps_per_s_two_streams": 2 * n / s_b, "ms_per_mixed_step_two_streams": s_b * 1e3,
            "ms_quadrotor_alone": s_q * 1e3, "ms_maze3d_alone": s_m * 1e3, "overlap_gain": (s_q + s_m) / s_b,
            "note": "one GPU's share of BASELINE configs[4]; the multi-GPU run is `bench.py --gpus N --workload mixed`"}
        del quad, maze
        torch.cuda.empty_cache()
    except Exception as e:
        out["mixed_error"] = repr(e)
    try:
        from metagym_amd.metalocomotion import MetaHumanoidEnv, variants
        n = 8192
        # the envs' default preset ("bullet"); fused auto-reset and a batch rolled to its STEADY STATE before
        # anything is timed: every variant starts with its feet in the ground and is thrown into the air by the first sub-step's
        # contact ERP, so the first ~150 steps after a reset are contact-free flight — timing those (rounds 1-3 did) flatters
        # the kernel. After 250 steps the batch is a mix of robots landing, lying, tumbling and restarting.
        env = MetaHumanoidEnv(num_envs=n, device=dev, auto_reset=True, max_steps=1000, seed=1)
        env.set_task(variants.models("humanoid", "TRAIN"))        # humanoid_var_tra_000..255, env e -> variant e % 256
        env.reset(seed=0)
        acts = [torch.rand(n, env.n_joints, device=dev) * 2 - 1 for _ in range(8)]
        s_flight = _time_steps(lambda i: env.step(acts[i % 8]), 12, 3)
        for i in range(250):
            env.step(acts[i % 8])
        on_ground = float((env.feet_contact.sum(0) > 0).float().mean())
        low = float((env.pos[2] < 1.0).float().mean())
        s = _time_steps(lambda i: env.step(acts[i % 8]), 24, 3)
        byt = 625                                                 # SURVEY.md §8(d) C4 bytes per env-step
        flop, flop_src, flop_mix = walker_flops("humanoid")
        if flop is None:
            flop, flop_src = 1.0e5, "SURVEY.md §8(d) estimate (no counted figure found under profiles/)"
        out["C4_humanoid_8192envs_256variants"] = {
            "env_steps_per_s": n / s, "ms_per_launch": s * 1e3, "preset": env.preset,
            "steady_state": {"preroll_steps": 265, "frac_envs_with_a_foot_on_the_ground": on_ground, "frac_torsos_below_1m": low,
                             "ms_per_launch_first_15_steps_after_reset_all_airborne": s_flight * 1e3},
            "roofline": {"bound": "valu", "achieved": flop * n / s / 1e12, "peak": FP64_VALU_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": flop * n / s / 1e12 / FP64_VALU_PEAK_TFLOPS,
                         "algorithmic_flop_per_env_step": flop, "flop_source": flop_src, "flop_mix_per_env_step": flop_mix,
                         "achieved_hbm_gbs": byt * n / s / 1e9, "hbm_frac": byt * n / s / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_env_step": byt,
                         "note": "f64 VALU peak; the flop figure is COUNTED (add + mul + 2 fma + div + sqrt + trig of the scalar C "
                                 "restatement of this kernel's algorithm, averaged over a rollout of this workload); one wave "
                                 "per env uses <= 29 of 64 lanes in every phase, so lane utilisation bounds this fraction at ~0.4"},
            "note": "physics parity unpinned (PyBullet is not in the reference tree)"}
        del env, acts
        torch.cuda.empty_cache()
        # ---- C4, contact-rich: the same 8 192 x 256-variant batch, but every robot starts STANDING on the floor
        #      (mjcf.grounded: the base lifted by the 9.3 cm the reference's variants start inside it) under U(-0.1, 0.1) actions and
        #      without auto-reset — the batch sags, kneels, falls and then lies on the ground with many proxies down. Three timings:
        #      the first 24 steps (standing -> kneeling), the lying steady state (after 200 steps), and the episodic auto-reset mix.
        from metagym_amd.metalocomotion import mjcf
        gmodels = [mjcf.grounded(m) for m in variants.models("humanoid", "TRAIN")]
        genv = MetaHumanoidEnv(num_envs=n, device=dev, auto_reset=False, max_steps=1000, seed=1)
        genv.set_task(gmodels)
        genv.reset(seed=0)
        gacts = [(torch.rand(n, genv.n_joints, device=dev) * 2 - 1) * 0.1 for _ in range(8)]
        s_stand = _time_steps(lambda i: genv.step(gacts[i % 8]), 24, 0)
        for i in range(176):
            genv.step(gacts[i % 8])
        s_lying = _time_steps(lambda i: genv.step(gacts[i % 8]), 24, 3)
        g_feet = float((genv.feet_contact.sum(0) > 0).float().mean())
        g_low = float((genv.pos[2] < 0.6).float().mean())
        g_bad = float(genv.bad_contacts.float().mean()) if getattr(genv, "bad_contacts", None) is not None else None
        g_flop, g_src, g_mix, g_rec = walker_flops("humanoid_lying", full=True)
        del genv
        torch.cuda.empty_cache()
        eenv = MetaHumanoidEnv(num_envs=n, device=dev, auto_reset=True, max_steps=1000, seed=1)
        eenv.set_task(gmodels)
        eenv.reset(seed=0)
        for i in range(120):                                      # three episode lengths (~44 steps each): a stationary mix
            eenv.step(gacts[i % 8])
        s_epi = _time_steps(lambda i: eenv.step(gacts[i % 8]), 24, 3)
        e_feet = float((eenv.feet_contact.sum(0) > 0).float().mean())
        e_flop, e_src, _, e_rec = walker_flops("humanoid_grounded", full=True)
        del eenv, gacts
        torch.cuda.empty_cache()
        fl = g_flop if g_flop is not None else flop
        out["C4_grounded_humanoid_8192envs_256variants"] = {
            "env_steps_per_s": n / s_lying, "ms_per_launch": s_lying * 1e3, "preset": "bullet",
            "workload": "C4's batch started standing on the floor (base lifted 9.3 cm), actions U(-0.1, 0.1), no auto-reset: timed "
                        "after 200 steps, every robot lying on the ground",
            "steady_state": {"preroll_steps": 203, "frac_envs_with_a_foot_on_the_ground": g_feet, "frac_torsos_below_0p6m": g_low,
                             "mean_non_foot_contact_points_per_env": g_bad,
                             "constraint_rows_per_substep_counted": None if g_rec is None else g_rec.get("constraint_rows_per_substep"),
                             "contacts_per_substep_counted": None if g_rec is None else g_rec.get("contacts_per_substep")},
            "ms_per_launch_first_24_steps_standing_to_kneeling": s_stand * 1e3,
            "episodic_autoreset": {"ms_per_launch": s_epi * 1e3, "env_steps_per_s": n / s_epi, "frac_envs_with_a_foot_on_the_ground": e_feet,
                                   "algorithmic_flop_per_env_step": e_flop, "flop_source": e_src,
                                   "constraint_rows_per_substep_counted": None if e_rec is None else e_rec.get("constraint_rows_per_substep"),
                                   "note": "same start, fused auto-reset: an episode ends when the torso sinks below 0.5 m (~44 steps), so "
                                           "most of the mix is standing / folding robots (round 6: Bullet's contact margin keeps their "
                                           "resting contacts — and the solver rows — alive between sub-steps)"},
            "roofline": {"bound": "valu", "achieved": fl * n / s_lying / 1e12, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": fl * n / s_lying / 1e12 / FP64_VALU_PEAK_TFLOPS, "algorithmic_flop_per_env_step": fl,
                         "flop_source": g_src if g_flop is not None else flop_src + " (the airborne rollout: no grounded count found)",
                         "flop_mix_per_env_step": g_mix, "achieved_hbm_gbs": byt * n / s_lying / 1e9,
                         "hbm_frac": byt * n / s_lying / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_env_step": byt},
            "contact_margin": "relative: 0.02 x the link's angular motion disc, Bullet's default rule (mjcf.contact_margins; round 6)",
            "note": "physics parity unpinned (PyBullet is not in the reference tree); under GPU parity against oracle/walker_oracle.c in "
                    "tests/test_walker_gpu.py::test_c4_grounded_configuration_sampled_against_oracle"}
    except Exception as e:
        out["walker_error"] = repr(e)
    try:
        # Quadrupedal: the A1 actuation sub-step on its own — ApplyAction + ReceiveObservation
        from metagym_amd.quadrupedal import A1Actuators
        n = ENVS_PER_GPU
        act = A1Actuators(n, dev)                                 # POSITION mode, control latency 0.002 s, pd latency 0
        f64 = dict(dtype=torch.float64, device=dev)
        q, qd = torch.rand(n, 12, **f64), torch.rand(n, 12, **f64)
        quat, rate, cmd = torch.rand(n, 4, **f64), torch.rand(n, 3, **f64), torch.rand(n, 12, **f64)
        qs, qds, quats, rates, cmds = [x.t().contiguous() for x in (q, qd, quat, rate, cmd)]
        act.Reset()
        act.ReceiveObservation(q, qd, quat, rate)

        def substep(i):       # one sub-step on SoA inputs: ReceiveObservation + the next ApplyAction, fused (one launch)
            rc = act._lib.mg_a1_receive_and_apply(ctypes.byref(act._cfg), n, ctypes.byref(act._st), qs.data_ptr(), qds.data_ptr(),
                                                  quats.data_ptr(), rates.data_ptr(), cmds.data_ptr(), None, 0.0,
                                                  act._torque.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
            assert rc == 0
        s = _time_steps(substep, 60, 10)
        # algorithmic bytes per robot sub-step (f64): apply: q, qd of the newest observation 192 + command 96 in, torque and
        # observed torque 192 out; receive: q, qd, quaternion, rate 248 + observed torque 96 in, one history entry 344 out,
        # control observation = blend of two entries 688 in, 344 out  ->  2 200 B
        byt = 192 + 96 + 192 + 248 + 96 + 344 + 688 + 344
        out["A1_actuation_substep_%denvs" % n] = {
            "robot_substeps_per_s": n / s, "us_per_substep_pair": s * 1e6,
            "roofline": {"bound": "hbm", "achieved": byt * n / s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": byt * n / s / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_robot_substep": byt},
            "note": "mg_a1_receive_and_apply (ReceiveObservation + the next sub-step's ApplyAction in one launch, four lanes per robot); bit-exact against the unmodified reference; "
                    "the actuation pair alone, no physics in this figure (quadrupedal_v0_urdf_* below runs the robot on the engine)"}
        del act
        torch.cuda.empty_cache()
        # the whole A1GymEnv.step composition (ETG + IK, 13 x (motor model, history), info, sensors, reward) around a NULL
        # physics that hands back constant joint states: the cost of everything the reference computes in Python per env step
        from metagym_amd.quadrupedal import A1GymEnv

        class NullPhysics(object):
            def __init__(self, n):
                from metagym_amd.quadrupedal import SoA
                self.s = tuple(SoA(x[:n].t().contiguous()) for x in (q, qd, quat, rate))  # [k][N], declared: taken without a copy
                self.w = dict(base=torch.zeros(n, 3, **f64), contact=torch.ones(n, 4, **f64),
                              bad=torch.zeros(n, dtype=torch.int32, device=dev))
            def reset(self, mask): return self.s
            def substep(self, torques): return self.s
            def world(self): return self.w
        n2 = 16384
        env = A1GymEnv(n2, NullPhysics(n2), dev, ETG=1, ETG_w=np.full((3, 20), 0.01), ETG_b=np.zeros(3))
        env.reset()
        a12 = torch.zeros(n2, 12, **f64)
        s2 = _time_steps(lambda i: env.step(a12), 20, 3)
        replay = env.capture_step()                  # the same step as one hipGraph
        env.reset()
        s2g = _time_steps(lambda i: replay(a12), 20, 3)
        out["A1GymEnv_python_side_%denvs_null_physics" % n2] = {
            "env_steps_per_s": n2 / s2, "ms_per_env_step": s2 * 1e3, "launches_per_env_step": 1 + 1 + 12 + 1 + 3,
            "ms_per_env_step_hipgraph": s2g * 1e3, "env_steps_per_s_hipgraph": n2 / s2g,
            "note": "ETG action path, 13 sub-steps of motor model + observation history, info, sensor stack, reward shaping — "
                    "what the reference's A1GymEnv.step computes in Python (3.3 ms per env step on one core of the build container, measured "
                    "with the same scripted world standing in for PyBullet, so ~3e2 env-steps/s/core), all on the GPU; "
                    "no physics in this figure"}
        del env
        torch.cuda.empty_cache()
        # quadrupedal-v0 WITH physics: a URDF robot on the articulated-body engine (metagym_amd.quadrupedal.A1Physics), 13 fused
        # sub-steps per launch with the PD motor model inside, 23 solver iterations. The file is the repo's A1-shaped demo
        # robot (examples/a1_like: NOT pybullet_data's a1.urdf, which is absent from the reference tree).
        import metagym_amd
        n3 = 8192
        urdf = os.path.join(ROOT, "examples", "a1_like", "a1_like.urdf")
        w = np.tile([[0.03], [0.0], [0.02]], (1, 20)) * np.sin(np.linspace(0, 2 * np.pi, 20))
        env = metagym_amd.make("quadrupedal-v0", num_envs=n3, urdf=urdf, device=dev, ETG=1, ETG_w=w, ETG_b=np.zeros(3), auto_reset=True)
        env.reset()
        a3 = torch.zeros(n3, 12, **f64)
        s3 = _time_steps(lambda i: env.step(a3), 20, 5)
        replay = env.capture_step()
        env.reset()
        s3g = _time_steps(lambda i: replay(a3), 20, 5)
        out["quadrupedal_v0_urdf_%denvs" % n3] = {
            "env_steps_per_s": n3 / s3, "ms_per_env_step": s3 * 1e3, "ms_per_env_step_hipgraph": s3g * 1e3,
            "env_steps_per_s_hipgraph": n3 / s3g, "physics_substeps_per_s_hipgraph": 13 * n3 / s3g,
            "env_steps_per_s_best": n3 / min(s3, s3g), "best_launch_mode": "eager" if s3 <= s3g else "hipgraph",
            "robot": "examples/a1_like/a1_like.urdf (13 bodies, 12 hinges, 124 contact proxies)",
            "note": "A1GymEnv.step end to end (ETG + IK, 13 x 2 ms engine sub-steps with the PD motor model inside one launch, "
                    "observation history, sensors, reward, fused per-robot reset); dynamics parity with PyBullet unpinned"}
        del env
        torch.cuda.empty_cache()
    except Exception as e:
        out["a1_error"] = repr(e)
    return out


# ---------------------------------------------------------------------------------------- in-run counter passes
PMC_PASSES = (("sq", ("SQ_INSTS_VALU", "SQ_WAVES")), ("fetch", ("FETCH_SIZE",)), ("write", ("WRITE_SIZE",)))


def pmc_child(dev, n, preroll):
    """`bench.py --pmc-child` (run under `rocprofv3 --pmc ...` by pmc_prepass): the headline workload — same shard plan,
    same steady-state pre-roll — and 24 eager launches of the step kernel; prints nothing."""
    q = QuadrotorShard(dev, shard_plan(0, 1, n, "quadrotor"), n, preroll=preroll)
    for i in range(24):
        q.step(i)
    torch.cuda.synchronize(dev)


def pmc_prepass(n, timeout_s=150):
    """Hardware counters of quadrotor_step_kernel observed IN THIS RUN: three `rocprofv3 --pmc` child runs of this script
    (separate passes — VALU instructions + waves, FETCH_SIZE, WRITE_SIZE — counters only, no trace domain, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes), averaged over the LAST 24 dispatches of each (the steady-state
    launches after the pre-roll). Returns {"valu_insts_per_wave", "hbm_bytes_per_launch", ...} with whatever passes worked,
    and "errors" for those that did not (no rocprofv3 on the box, a timeout, ...): never raises."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    out, errors = {}, {}
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"errors": {"rocprofv3": "not found"}}
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "BENCH_FORCE_DIST"):
        env.pop(k, None)
    for tag, counters in PMC_PASSES:
        d = tempfile.mkdtemp(prefix="mg_pmc_%s_" % tag, dir="/tmp")
        cmd = [exe, "--pmc"] + list(counters) + ["--output-format", "csv", "-d", d, "-o", "q", "--", sys.executable,
                                                  os.path.abspath(__file__), "--pmc-child", "--envs-per-gpu", str(n)]
        try:
            t0 = time.perf_counter()
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
            vals = {}
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(path)):
                    if "quadrotor_step_kernel" in r["Kernel_Name"]:
                        vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            for c in counters:
                if not vals.get(c):
                    raise RuntimeError("no %s rows for quadrotor_step_kernel" % c)
                out[c] = sum(vals[c][-24:]) / len(vals[c][-24:])
            out.setdefault("pass_seconds", {})[tag] = time.perf_counter() - t0
        except Exception as e:
            errors[tag] = repr(e)[:200]
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if "SQ_INSTS_VALU" in out and out.get("SQ_WAVES"):
        out["valu_insts_per_wave"] = out["SQ_INSTS_VALU"] / out["SQ_WAVES"]
    if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
        # both counters are in KB; on gfx950 FETCH_SIZE reports half the bytes of a coalesced read stream (the guide's correction)
        out["hbm_bytes_per_launch"] = (2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0
    if errors:
        out["errors"] = errors
    return out


# ---------------------------------------------------------------------------------------- multi-GPU plumbing
def gather_walls(dist, wall):
    """Every rank's wall time of the timed region (CPU tensors over gloo), for the record."""
    if dist is None:
        return [wall]
    ws = [torch.zeros(1, dtype=torch.float64) for _ in range(dist.get_world_size())]
    dist.all_gather(ws, torch.tensor([wall], dtype=torch.float64))
    return [float(w.item()) for w in ws]


def aggregate_throughput(dist, wall, envs_per_rank, steps):
    """Whole-job env-steps/s: every rank stepped `envs_per_rank` envs `steps` times; the job took as
    long as its slowest rank. `dist` is torch.distributed (initialised, any backend) or None for one process.
    The reduction runs on a CPU tensor (gloo): nothing here touches RCCL."""
    wall_max, world = wall, 1
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall_max = float(t.item())
        world = dist.get_world_size()
    return world * envs_per_rank * steps / wall_max, wall_max


def solo_wall_of(dist, rank, run):
    """--self-baseline: `run()` (the K timed steps) on rank 0 while every other rank waits at a barrier — the N = 1 figure of
    this very job (same box, process and launch mode), so the scaling efficiency needs no second invocation. Returns rank 0's
    wall time on rank 0, None elsewhere."""
    if dist is not None:
        dist.barrier()
    w = None
    if rank == 0:
        t0 = time.perf_counter()
        run()
        w = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    return w


def scaling_fields(value, world, envs_per_rank, steps, rank_walls, solo_wall=None, single_gpu_value=None):
    """What the line says about scaling, as plain numbers (rank 0): every rank's wall time of the timed region and its own
    throughput over THAT time (`value` is the job's: all envs over the slowest rank's wall); with --self-baseline the solo
    figure and efficiency = value(N) / (N x solo); with --single-gpu-value V the same against a prior N = 1 run."""
    out = {"rank_wall_ms": [w * 1e3 for w in rank_walls],
           "rank_value": [envs_per_rank * steps / w for w in rank_walls]}
    if solo_wall is not None:
        solo = envs_per_rank * steps / solo_wall
        out["self_baseline"] = {"value": solo, "ms_per_step": solo_wall / steps * 1e3,
                                "what": "rank 0 alone, same process / launch mode / K steps, the other ranks idle at a barrier"}
        out["efficiency"] = value / (world * solo)
    if single_gpu_value:
        out["efficiency" if solo_wall is None else "efficiency_vs_single_gpu_value"] = value / (world * single_gpu_value)
        out["single_gpu_value"] = single_gpu_value
    return out


def shard_env_ids(rank, world, envs_per_rank):
    """Global env ids owned by `rank` (contiguous shards)."""
    return np.arange(rank * envs_per_rank, (rank + 1) * envs_per_rank, dtype=np.int64)


def shard_plan(rank, world, envs_per_rank, workload="quadrotor", job_seed=1000, n_maze_tasks=64):
    """Everything rank `rank` needs to build its share of the job, as plain numbers (exercised on CPU by
    tests/test_distributed_cpu.py): the global ids of its envs, the `env_id_base` handed to the fused auto-reset
    (Philox streams are keyed by GLOBAL env id, so the union over ranks is the single-rank job), the seed of its
    host-side initial reset and action streams, and for the mixed workload the maze task of every env
    (global env id mod table size — the table itself is replicated)."""
    ids = shard_env_ids(rank, world, envs_per_rank)
    plan = {"rank": rank, "world": world, "env_ids": ids, "env_id_base": int(ids[0]), "job_seed": job_seed,
            "reset_seed": job_seed + rank, "action_seed": 2 * job_seed + rank}
    if workload == "mixed":
        plan["maze_env_ids"] = ids
        plan["maze_task_ids"] = (ids % n_maze_tasks).astype(np.int32)
    return plan


class QuadrotorShard:
    """C2 on one GPU: this rank's quadrotors + resident action batches."""

    def __init__(self, dev, plan, n, preroll=None, nt=1000):
        import metagym_amd
        self.n = n
        self.env = metagym_amd.make("quadrotor-v0", num_envs=n, device=dev, task="hovering_control", nt=nt,
                                    auto_reset=True, seed=plan["job_seed"], env_id_base=plan["env_id_base"])
        self.env.reset(seed=plan["reset_seed"])
        g = torch.Generator(device=dev)
        g.manual_seed(plan["action_seed"])
        self.actions = torch.rand(N_ACTION_BATCHES, n, 4, device=dev, generator=g) * 14.9 + 0.1
        self.action_list = [self.actions[i] for i in range(N_ACTION_BATCHES)]     # views made once: indexing costs ~1.2 us per step
        # Steady state before anything is timed: a fresh batch has every env at ct = 0, so a short timed region
        # would contain no episode end at all (and none of the fused in-launch reset work). Episode clocks start
        # staggered over [0, nt) by GLOBAL env id (shard-invariant) and the batch is rolled `preroll` untimed
        # steps (default nt: every env has ended at least once, by the clock or on the floor).
        nt = int(self.env.nt)
        ids = torch.arange(n, device=dev, dtype=torch.int64) + int(plan["env_id_base"])
        sd = self.env.state_dict()
        sd["ct"] = ((ids * 977) % nt).to(torch.int32)
        self.env.load_state_dict(sd)
        self.preroll_steps = nt if preroll is None else int(preroll)
        for i in range(self.preroll_steps):
            self.step(i)
        torch.cuda.synchronize(dev)

    def step(self, i):
        self.env.step(self.action_list[i % N_ACTION_BATCHES])

    def episodes(self):
        """Total number of in-launch restarts so far (sum of the per-env episode counters)."""
        return int(self.env.episode.to(torch.int64).sum().item())


class MazeShard:
    """C5's MetaMaze3D half on one GPU: 9x9 mazes, 64x64 int32 frames, SURVIVAL, fused auto-reset."""

    def __init__(self, dev, plan, n, res=64, max_steps=200):
        import metagym_amd
        self.n = n
        self.env = metagym_amd.make("meta-maze-discrete-3D-v0", num_envs=n, device=dev, max_steps=max_steps,
                                    resolution=(res, res), task_type="SURVIVAL", auto_reset=True)
        self.env.set_task(maze_tasks(), task_ids=torch.as_tensor(plan["maze_task_ids"]))
        self.env.reset()
        g = torch.Generator(device=dev)
        g.manual_seed(plan["action_seed"] + 7)
        self.actions = [torch.randint(0, 4, (n,), device=dev, dtype=torch.int32, generator=g) for _ in range(4)]

    def step(self, i):
        self.env.step(self.actions[i % 4])


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _spawned_rank(local_rank, argv, world, port):
    """One rank of a bare `python bench.py --gpus N` (no torch.distributed.run around it): the environment
    torch.distributed.run would have provided, then the ordinary main()."""
    os.environ.update({"RANK": str(local_rank), "LOCAL_RANK": str(local_rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "BENCH_SPAWNED": "1"})
    main(argv)


def self_spawn(argv, world):
    """`python bench.py --gpus N` without a launcher: fork-free spawn of N ranks on this node (one process per GPU,
    same code path as under torch.distributed.run); rank 0 prints the one JSON line."""
    import torch.multiprocessing as mp
    mp.spawn(_spawned_rank, args=(list(argv), world, _free_port()), nprocs=world, join=True)


class MixedStep:
    """C5's step on one GPU: the two families are independent, so the quadrotor launch and the maze3d launch go to two
    HIP streams and may run concurrently (tests/test_mixed_gpu.py: identical, bit for bit, to each family stepped alone)."""

    def __init__(self, dev, quad, maze):
        self.quad, self.maze = quad, maze
        self.st_q, self.st_m = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

    def __call__(self, i):
        with torch.cuda.stream(self.st_q):
            self.quad.step(i)
        with torch.cuda.stream(self.st_m):
            self.maze.step(i)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--workload", choices=("quadrotor", "mixed"), default="quadrotor")
    ap.add_argument("--launch", choices=("auto", "graph", "eager"), default="auto",
                    help="auto (default): the K timed steps are run twice, replayed as one hipGraph and as K eager launches, each in its "
                         "own barrier-bracketed region; the mode with the higher whole-job throughput is the line's value, the other "
                         "is recorded under sanity.other_launch_mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C1/C3/C4 side measurements")
    ap.add_argument("--preroll", type=int, default=None,
                    help="untimed steps that bring the batch to its steady episode-age mix (default: nt = 1000)")
    ap.add_argument("--single-gpu-value", type=float, default=None,
                    help="a prior N=1 `value`; with it the line carries efficiency = value(N) / (N * value(1))")
    ap.add_argument("--self-baseline", action="store_true",
                    help="with N > 1 ranks: rank 0 first times the same K steps ALONE (the other ranks idle at a barrier), so the line "
                         "carries efficiency = value(N) / (N * solo value) without a prior --single-gpu-value run")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 --pmc passes (VALU instruction count, HBM traffic)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--plan-only", action="store_true",
                    help="rendezvous + shard plans only, on CPU (what tests/test_distributed_cpu.py drives); no GPU work")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_spawn(argv, args.gpus)     # bare `python bench.py --gpus N`: spawn the N ranks ourselves
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not args.plan_only:
        have = torch.cuda.device_count()
        if have <= local_rank or (args.gpus > 1 and have < min(args.gpus, int(os.environ.get("LOCAL_WORLD_SIZE", args.gpus)))):
            raise SystemExit("bench.py --gpus %d: rank %d (local rank %d) needs its own GPU but torch.cuda.device_count() == %d"
                             % (args.gpus, rank, local_rank, have))
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":   # the env var lets a 1-GPU box exercise this path
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # gloo's C++ side reports its peer count on STDOUT ("[Gloo] Rank 0 is connected to ..."): keep the one JSON line
        # this script owes its caller alone there by pointing fd 1 at stderr while the context comes up
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo", rank=rank, world_size=world)   # CPU barrier only: no RCCL in this harness
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    def barrier():
        if dist is not None:
            dist.barrier()

    n = args.envs_per_gpu
    mixed = args.workload == "mixed"
    plan = shard_plan(rank, world, n, args.workload)
    if args.plan_only:
        # every rank reports its shard; rank 0 checks that the union is the single-process job and prints it
        lo_hi = torch.tensor([int(plan["env_ids"][0]), int(plan["env_ids"][-1]) + 1, plan["env_id_base"]], dtype=torch.int64)
        gathered = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
        if dist is not None:
            dist.all_gather(gathered, lo_hi)
        else:
            gathered = [lo_hi]
        if rank == 0:
            shards = [[int(x) for x in g] for g in gathered]
            ok = all(shards[r][0] == r * n and shards[r][1] == (r + 1) * n and shards[r][2] == r * n for r in range(world))
            print(json.dumps({"plan_only": True, "n_gpus": world, "envs_per_gpu": n, "shards": shards,
                              "union_is_single_job": bool(ok), "spawned": os.environ.get("BENCH_SPAWNED") == "1"}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return
    if args.pmc_child:
        return pmc_child(dev, n, args.preroll)
    quad = QuadrotorShard(dev, plan, n, preroll=args.preroll)
    maze = MazeShard(dev, plan, n) if mixed else None
    step = MixedStep(dev, quad, maze) if mixed else quad.step

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    auto = args.launch == "auto" and not mixed
    launch_mode = ("graph" if auto else args.launch) if not mixed else "eager"
    graph = None
    if launch_mode == "graph":
        try:
            graph = capture_steps(dev, step, args.steps)
            graph.replay()            # first replay uploads the executable graph: keep that out of the timed region
            torch.cuda.synchronize(dev)
        except Exception as e:        # report, fall back to the eager loop (same launches)
            launch_mode = "eager (graph capture failed: %r)" % (e,)
            graph = None

    def timed_region():
        # torch creates the HIP event behind a torch.cuda.Event at its FIRST record(): do that before the clock starts (the two
        # lazy hipEventCreate calls cost ~0.1 ms of host time — a third of a 20-step timed region)
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record()
        ev1.record()
        # the W untimed warm-up steps come IMMEDIATELY before the bracket (round 5): a timed region that starts after host-side
        # bookkeeping (graph capture, episode counters) finds a GPU that has idled for milliseconds, and a 20-step region — 0.3 ms —
        # then pays the clock ramp on every one of its launches (17.1 us per step single-shot against 15.7 us for the same 20
        # eager steps repeated back to back, profiles/r05/quad_host_cost.txt)
        for i in range(args.warmup):
            step(i)
        barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ev0.record()                  # torch's current stream == the stream the kernels / the graph are launched on
        if graph is not None:
            graph.replay()
        else:
            for i in range(args.steps):
                step(i)
        ev1.record()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        barrier()
        return t1 - t0, ev0.elapsed_time(ev1)

    def run_k_steps():
        torch.cuda.synchronize(dev)
        if graph is not None:
            graph.replay()
        else:
            for i in range(args.steps):
                step(i)
        torch.cuda.synchronize(dev)

    solo_wall = solo_wall_of(dist, rank, run_k_steps) if (args.self_baseline and world > 1) else None
    ep0 = quad.episodes()
    wall, dev_ms = timed_region()
    ep1 = quad.episodes()
    envs_per_rank = n * (2 if mixed else 1)
    value, wall_max = aggregate_throughput(dist, wall, envs_per_rank, args.steps)
    rank_walls = gather_walls(dist, wall)

    # the other launch mode: a second timed region of the same K steps. `--launch auto`: whichever mode gives the higher whole-job
    # throughput (max-over-ranks wall, so every rank decides alike) is the line's value; an explicit --launch: for the record only
    other = None
    if not mixed and graph is not None and (auto or world == 1):
        graph_keep, graph = graph, None
        e0 = quad.episodes()
        w2, d2 = timed_region()               # eager
        e1 = quad.episodes()
        graph = graph_keep
        v2, wmax2 = aggregate_throughput(dist, w2, envs_per_rank, args.steps) if auto else (0.0, 0.0)
        rw2 = gather_walls(dist, w2) if auto else None
        this = {"mode": "graph", "host_wall_ms_per_step": wall / args.steps * 1e3, "hip_event_us_per_step": dev_ms / args.steps * 1e3}
        that = {"mode": "eager", "host_wall_ms_per_step": w2 / args.steps * 1e3, "hip_event_us_per_step": d2 / args.steps * 1e3}
        if auto and v2 > value:
            value, wall_max, rank_walls, wall, dev_ms, ep0, ep1 = v2, wmax2, rw2, w2, d2, e0, e1
            launch_mode, graph, other = "eager", None, this
        else:
            other = that
        if auto:
            other["chosen_by"] = "--launch auto: higher whole-job throughput of the two timed regions"

    # What a short timed region pays on top of its kernels: the same env.step() launches as hipGraphs of K and
    # of 10 K steps, each replayed a few times between synchronisations; T(K) = a + b K gives the per-step cost b of a long run and
    # the fixed cost a of ONE timed region (graph launch, first-dispatch latency after an idle queue, completion wake-up).
    graph_fit = None
    if not mixed and world == 1 and args.launch != "eager":
        try:
            def replay_wall(g, reps=5):
                best_w, best_d = 1e9, 1e9
                for _ in range(reps):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); e1.record()
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    e0.record()
                    g.replay()
                    e1.record()
                    torch.cuda.synchronize(dev)
                    best_w, best_d = min(best_w, time.perf_counter() - t0), min(best_d, e0.elapsed_time(e1) * 1e-3)
                return best_w, best_d
            k1, k2 = args.steps, min(10 * args.steps, 400)
            g1 = graph if graph is not None else capture_steps(dev, step, k1)
            g2 = capture_steps(dev, step, k2)
            g2.replay()
            torch.cuda.synchronize(dev)
            (w1, d1), (w2, d2) = replay_wall(g1), replay_wall(g2)
            b_w, b_d = (w2 - w1) / (k2 - k1), (d2 - d1) / (k2 - k1)
            graph_fit = {"k": [k1, k2], "host_wall_us": [w1 * 1e6, w2 * 1e6], "hip_event_us": [d1 * 1e6, d2 * 1e6],
                         "per_step_us_host_wall": b_w * 1e6, "per_step_us_hip_events": b_d * 1e6,
                         "fixed_us_per_timed_region_host_wall": (w1 - b_w * k1) * 1e6,
                         "fixed_us_per_timed_region_hip_events": (d1 - b_d * k1) * 1e6,
                         "note": "best of 5 replays each; T(K) = fixed + per_step x K. ms_per_step of a K-step region = per_step + fixed / K"}
            del g2
        except Exception as e:
            graph_fit = {"error": repr(e)}
    done_frac = float(quad.env._done.float().mean().item())
    failed_any = int(quad.env._failed.max().item())
    if rank == 0:
        launch_s = dev_ms * 1e-3 / args.steps
        out = {
            "metric": METRIC,
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
            # what "the reference's arithmetic" means here, said in the line
            "numerics": "every expression in the dtype NumPy 2.2.6 (NEP 50) evaluates it in — the only way the reference runs offline, "
                        "and what every bit-exact statement of this repo is against. The reference pins numpy 1.22, where a python float "
                        "times a float32 scalar is float64: measured gap between the two readings <= 4.1e-6 over 1 000 Euler sub-steps, "
                        "1.5e-5 (state) / 8.3e-5 (Euler-angle observation) over tumbling 400 - 1 000-step rollouts "
                        "(profiles/r03/numpy_pin_gap.txt); the kernel has no numpy-1.22 mode",
            "data": "synthetic",
        }
        out.update(scaling_fields(value, world, envs_per_rank, args.steps, rank_walls, solo_wall, args.single_gpu_value))
        if not mixed:
            achieved = BYTES_PER_ENV_STEP * n / launch_s / 1e9
            out["config"] = {"workload": "Quadrotor hovering_control, %d envs/GPU, dt=0.01 (10 Euler sub-steps), "
                                         "nt=1000, flat map, actions U(0.1,15) f32, fused auto-reset" % n,
                             "launch": "one quadrotor_step_kernel launch per env.step(); %s" % (
                                 "the %d timed env.step() calls replayed as one hipGraph" % args.steps
                                 if graph is not None else ("the %d timed env.step() calls issued one by one from Python (%s)"
                                                            % (args.steps, launch_mode))),
                             "envs_per_gpu": n, "sharding": "env-sharded, no collective; gloo barrier for timing only"}
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_ceiling": achieved / COPY_CEILING_GBS,
                               "traffic": None,
                               "kernel": "quadrotor_step_kernel", "avg_launch_us": launch_s * 1e6,
                               "algorithmic_bytes_per_launch": BYTES_PER_ENV_STEP * n,
                               "note": "achieved / peak / frac: ALGORITHMIC bytes (317 B/env-step, SURVEY.md §8d) over the average "
                                       "launch duration (HIP events around the timed region / steps, inter-launch gaps included) "
                                       "against the 8 TB/s HBM peak. The kernel is NOT HBM-bound: `valu_issue` is the fraction of "
                                       "the VALU issue rate (one wave-instruction per 4 clk per SIMD, 1024 SIMDs at 2.4 GHz) its "
                                       "measured instruction count reaches in that same duration; `bound` names the larger"}
            pmc = {} if (args.no_pmc or world != 1) else pmc_prepass(n)
            if "hbm_bytes_per_launch" in pmc:
                out["roofline"]["traffic"] = pmc["hbm_bytes_per_launch"]
                out["roofline"]["traffic_over_algorithmic"] = pmc["hbm_bytes_per_launch"] / float(BYTES_PER_ENV_STEP * n)
                out["roofline"]["traffic_source"] = ("this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate child passes of the same "
                                                     "workload, (2 x FETCH_SIZE + WRITE_SIZE) KB per launch, last 24 launches")
            if "valu_insts_per_wave" in pmc:
                waves = (n + 63) // 64
                peak_issue = 1024 * 2.4e9 / 4.0
                ach = pmc["valu_insts_per_wave"] * waves / launch_s
                out["roofline"]["valu_issue"] = {"achieved": ach, "peak": peak_issue, "unit": "wave-instr/s", "frac": ach / peak_issue,
                                                 "valu_insts_per_wave": pmc["valu_insts_per_wave"], "waves_per_launch": waves,
                                                 "source": "this run: rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES child pass of the same "
                                                           "workload (last 24 launches), launch duration from the timed region"}
                # round 6: "one instruction per 4 clk" is the nominal rate; the measured issue costs on gfx950 (scripts/ubench/f64_rates.hip:
                # f64 FMA / MUL 5.4 - 5.7 clk, conversions 4.3, f32 2.3) weighted by this kernel's measured class mix
                # (profiles/r06/valu_mix.json) give the pipe's real utilisation
                try:
                    vm = json.load(open(os.path.join(ROOT, "profiles", "r06", "valu_mix.json")))["kernels"]["quad"]
                    cyc = float(vm["mean_issue_cycles_per_valu_instruction"])
                    out["roofline"]["valu_issue"]["mean_issue_cycles_per_instruction_measured_mix"] = cyc
                    out["roofline"]["valu_issue"]["pipe_utilisation_at_measured_costs"] = ach * cyc / (1024 * 2.4e9)
                    out["roofline"]["valu_issue"]["mix_source"] = "profiles/r06/valu_mix.json (class counters) x profiles/r06/valu_issue_rates_gfx950.txt"
                except Exception:
                    pass
                if ach / peak_issue > out["roofline"]["frac"]:
                    out["roofline"]["bound"] = "valu_issue"
                out["roofline"]["bound_frac"] = max(ach / peak_issue, out["roofline"]["frac"])
            if pmc.get("errors") or not pmc:
                out["roofline"]["pmc_passes"] = {"skipped": bool(args.no_pmc or world != 1), "errors": pmc.get("errors")}
            elif "pass_seconds" in pmc:
                out["roofline"]["pmc_pass_seconds"] = pmc["pass_seconds"]
        else:
            frame_b = 12 * 64 * 64 + 64
            byt = (BYTES_PER_ENV_STEP + frame_b) * n
            out["config"] = {"workload": "mixed (BASELINE configs[4]): %d Quadrotor hovering_control + %d "
                                         "MetaMazeDiscrete3D 9x9 @64x64 envs per GPU, %d envs in the job"
                                         % (n, n, 2 * n * world),
                             "launch": "one quadrotor launch + one maze3d launch per step, two HIP streams, eager",
                             "envs_per_gpu": 2 * n, "sharding": "env-sharded, no collective; gloo barrier for timing only"}
            out["roofline"] = {"bound": "hbm", "achieved": byt / (wall / args.steps) / 1e9, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": byt / (wall / args.steps) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                               "kernel": "quadrotor_step_kernel + maze3d_step_kernel (concurrent)",
                               "algorithmic_bytes_per_step": byt,
                               "note": "rank 0's wall time per mixed step; per-kernel rooflines are in the "
                                       "single-workload run"}
        out["sanity"] = {"done_frac_last_step": done_frac, "failed_max": failed_any,
                         # ep0 / ep1 bracket timed_region(), which runs the W warm-up steps and then the K timed ones
                         "episode_ends_in_warmup_plus_timed_region": ep1 - ep0,
                         "done_frac_timed_region": (ep1 - ep0) / float(n * (args.steps + args.warmup)),
                         "preroll_steps": quad.preroll_steps,
                         "host_wall_ms_per_step": wall / args.steps * 1e3, "launch_mode": launch_mode,
                         "other_launch_mode": other, "graph_launch_fit": graph_fit}
        if graph_fit and "fixed_us_per_timed_region_host_wall" in graph_fit:
            out["graph_launch_us"] = graph_fit["fixed_us_per_timed_region_host_wall"]
            out["kernel_us_per_step_long_run"] = graph_fit["per_step_us_hip_events"]
        if not mixed:
            out["from_profiles"] = from_profiles(n, launch_s)
        if world == 1 and not args.no_secondary and not mixed:
            del quad
            torch.cuda.empty_cache()
            ipw_run = out.get("roofline", {}).get("valu_issue", {}).get("valu_insts_per_wave")
            out["secondary"] = secondary_workloads(dev, valu_insts_per_wave=ipw_run)
            # the other BASELINE configs as short top-level scalars (details under `secondary`)
            sec = out["secondary"]

            def pick(key, *path):
                v = sec.get(key)
                for k in path:
                    v = v.get(k) if isinstance(v, dict) else None
                return v if isinstance(v, (int, float)) else None
            for name, key, path in (
                    ("C3_maze3d_discrete_env_steps_per_s", "C3_maze3d_discrete_9x9_256x256_16384envs", ("env_steps_per_s",)),
                    ("C3_maze3d_discrete_hbm_frac", "C3_maze3d_discrete_9x9_256x256_16384envs", ("roofline", "frac")),
                    ("C3_maze3d_continuous_env_steps_per_s", "C3_maze3d_continuous_9x9_256x256_16384envs", ("env_steps_per_s",)),
                    ("C3_maze3d_continuous_hbm_frac", "C3_maze3d_continuous_9x9_256x256_16384envs", ("roofline", "frac")),
                    ("C4_humanoid_env_steps_per_s", "C4_humanoid_8192envs_256variants", ("env_steps_per_s",)),
                    ("C4_humanoid_f64_valu_frac", "C4_humanoid_8192envs_256variants", ("roofline", "frac")),
                    ("C4_grounded_env_steps_per_s", "C4_grounded_humanoid_8192envs_256variants", ("env_steps_per_s",)),
                    ("C4_grounded_f64_valu_frac", "C4_grounded_humanoid_8192envs_256variants", ("roofline", "frac")),
                    ("C5_share_env_steps_per_s", "C5_mixed_share_65536quad_plus_65536maze3d_64x64", ("env_steps_per_s_two_streams",)),
                    ("C1_maze2d_1env_us_per_step_hipgraph", "C1_maze2d_15x15_escape_1env", ("us_per_step_hipgraph_100",)),
                    ("north_star_2p17_env_steps_per_s", "north_star_quadrotor_hovering_131072envs_1gpu", ("env_steps_per_s",)),
                    ("north_star_2p20_env_steps_per_s", "north_star_quadrotor_hovering_1048576envs_1gpu", ("env_steps_per_s",)),
                    ("quadrupedal_v0_env_steps_per_s", "quadrupedal_v0_urdf_8192envs", ("env_steps_per_s_best",))):
                v = pick(key, *path)
                if v is not None:
                    out[name] = v
        if world == 1 and not args.no_cpu_baseline:
            port = cpu_port()
            ref, why = cpu_reference()
            if ref is not None:
                out["cpu_baseline"] = ref
                out["cpu_port"] = port
            else:
                # say it in the baseline itself: this is the C port, and the reference's own path was not timed on this box
                port["reference_timed_here"] = False
                port["reference_unavailable"] = why
                ref_fig = out.get("from_profiles", {}).get("reference_cpu")
                if ref_fig:
                    port["reference_figure_other_machine"] = ref_fig
                out["cpu_baseline"] = port
                out["reference_unavailable"] = why
            sec = out.get("secondary", {})
            for key, fn in (("C3_maze3d_discrete_9x9_256x256_16384envs", cpu_baseline_maze3d),
                            ("C4_humanoid_8192envs_256variants", cpu_baseline_walker)):
                if key in sec:
                    try:
                        sec[key]["cpu_baseline"] = fn()
                    except Exception as e:
                        sec[key]["cpu_baseline_error"] = repr(e)
            if "C1_maze2d_15x15_escape_1env" in sec:
                r1 = cpu_reference_maze2d()
                if r1 is not None:
                    sec["C1_maze2d_15x15_escape_1env"]["cpu_reference"] = r1
                    if "us_per_step" in r1.get("ESCAPE", {}):
                        out["C1_reference_cpu_us_per_step"] = r1["ESCAPE"]["us_per_step"]
                    if "us_per_step" in r1.get("SURVIVAL", {}):
                        out["C1_reference_cpu_us_per_step_survival"] = r1["SURVIVAL"]["us_per_step"]
            if "C3_maze3d_discrete_9x9_256x256_16384envs" in sec:
                r3 = cpu_reference_maze3d()
                if r3 is not None:
                    sec["C3_maze3d_discrete_9x9_256x256_16384envs"]["cpu_reference_unjitted"] = r3
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
