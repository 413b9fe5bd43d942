"""MetaMaze golden vectors (called from gen_golden.py; TEST INFRASTRUCTURE, build container only).

Runs the unmodified reference envs (metagym/metamaze/envs/maze_env.py:16,85,155) through the
shims in refstubs/ and records, per step: agent grid / heading / location, life, reward (f64), done,
steps and the observation (2-D: float32 window maze_2d.py:89-121; 3-D: int32 image
ray_caster_utils.py:66-209 + life bar maze_discrete_3d.py:118-126). After `done` the reference
demands reset() (maze_env.py:60-61); the generator resets and keeps going on the same task, and
records where it did so.

Files
  maze_textures.npz          the 7 64x64 wall/ground textures + ceiling the reference loads
                             (maze_task.py:19-36), as uint8 — inputs of the renderer
  maze2d_{task}_s{seed}.npz  MetaMaze2D, n=15, view_grid=1 (BASELINE config C1) and view_grid=2
  maze3d_disc_*.npz          MetaMazeDiscrete3D, n=9 (config C3): 64x64, 48x32 and a few 256x256 frames
  maze3d_cont_*.npz          MetaMazeContinuous3D, n=9, 64x64
"""
import os
import random

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def _task_dict(task):
    return dict(
        start=np.asarray(task.start, np.int32), goal=np.asarray(task.goal, np.int32),
        cell_walls=np.asarray(task.cell_walls, np.int32), cell_texts=np.asarray(task.cell_texts, np.int32),
        cell_size=np.float64(task.cell_size), wall_height=np.float64(task.wall_height),
        agent_height=np.float64(task.agent_height), initial_life=np.float64(task.initial_life),
        max_life=np.float64(task.max_life), step_reward=np.float64(task.step_reward),
        goal_reward=np.float64(task.goal_reward), food_rewards=np.asarray(task.food_rewards, np.float64),
        food_interval=np.asarray(task.food_interval, np.int32))


def _sample_task(seed, **kw):
    from metagym.metamaze import MazeTaskSampler
    random.seed(seed)
    np.random.seed(seed)
    return MazeTaskSampler(**kw)


def _run(env, task, actions, kind, record_obs_every=1):
    core = env.maze_core
    env.set_task(task)
    obs0 = env.reset()
    rec = {k: [] for k in ("grid", "reward", "done", "steps", "life", "obs", "obs_step", "reset_before",
                           "ori_idx", "ori", "loc")}
    need_reset = False
    for t, a in enumerate(actions):
        rec["reset_before"].append(need_reset)
        if need_reset:
            env.reset()
            need_reset = False
        if kind == "cont":
            obs, reward, done, info = env.step((float(a[0]), float(a[1])))
        else:
            obs, reward, done, info = env.step(int(a))
        rec["grid"].append(np.asarray(core._agent_grid, np.int32))
        rec["reward"].append(np.float64(reward))
        rec["done"].append(bool(done))
        rec["steps"].append(int(info["steps"]))
        rec["life"].append(np.float64(getattr(core, "_life", 0.0)))
        rec["ori_idx"].append(int(getattr(core, "_agent_ori_index", 0)))
        rec["ori"].append(np.float64(core._agent_ori))
        rec["loc"].append(np.asarray(core._agent_loc, np.float64))
        if t % record_obs_every == 0 or done:
            rec["obs"].append(np.asarray(obs))
            rec["obs_step"].append(t)
        need_reset = bool(done)
    out = {k: np.asarray(v) for k, v in rec.items()}
    out["obs0"] = np.asarray(obs0)
    out["actions"] = np.asarray(actions)
    out["numpy_version"] = np.str_(np.__version__)
    for k, v in _task_dict(task).items():
        out["task_" + k] = v
    return out


def gen_maze(gym):
    from metagym.metamaze.envs.maze_task import MAZE_TASK_MANAGER
    grounds = np.asarray(MAZE_TASK_MANAGER.grounds)
    assert grounds.dtype == np.float32 and np.array_equal(grounds, np.round(grounds))
    np.savez_compressed(os.path.join(OUT, "maze_textures.npz"), grounds=grounds.astype(np.uint8),
                        ceil=np.asarray(MAZE_TASK_MANAGER.ceil, np.uint8))
    print("wrote maze_textures.npz", grounds.shape, MAZE_TASK_MANAGER.ceil.shape)

    # ---- MetaMaze2D (config C1: 15x15, view_grid=1) -------------------------------------------
    for task_type in ("ESCAPE", "SURVIVAL"):
        for seed, view_grid, food_density in ((0, 1, 0.010), (1, 1, 0.05), (2, 2, 0.05)):
            task = _sample_task(seed, n=15, allow_loops=True, crowd_ratio=0.35, step_reward=-0.01,
                                goal_reward=1.0, food_density=food_density, food_interval=7)
            env = gym.make("meta-maze-2D-v0", max_steps=60, enable_render=False, view_grid=view_grid,
                           task_type=task_type)
            actions = np.random.RandomState(seed + 1).randint(0, 4, size=300)
            d = _run(env, task, actions, "2d")
            d["view_grid"] = np.int32(view_grid)
            d["max_steps"] = np.int32(60)
            name = "maze2d_%s_s%d.npz" % (task_type.lower(), seed)
            np.savez_compressed(os.path.join(OUT, name), **d)
            print("wrote", name, "dones", int(d["done"].sum()), "reward sum %.3f" % d["reward"].sum())

    # ---- MetaMazeDiscrete3D (config C3: 9x9) ---------------------------------------------------
    specs = [("ESCAPE", 0, (64, 64), 90, 1), ("SURVIVAL", 1, (64, 64), 90, 1), ("SURVIVAL", 2, (48, 32), 60, 1),
             ("ESCAPE", 3, (256, 256), 8, 1), ("SURVIVAL", 4, (256, 256), 8, 1)]
    for task_type, seed, res, T, every in specs:
        task = _sample_task(seed, n=9, allow_loops=False, step_reward=-0.01, goal_reward=1.0,
                            food_density=0.06, food_interval=5)
        env = gym.make("meta-maze-discrete-3D-v0", max_steps=40, enable_render=False, task_type=task_type,
                       resolution=res)
        # biased towards "forward" so the agent actually travels; all four actions occur
        actions = np.random.RandomState(seed + 1).choice(4, size=T, p=[0.2, 0.2, 0.1, 0.5])
        d = _run(env, task, actions, "disc", every)
        d["resolution"] = np.asarray(res, np.int32)
        d["max_steps"] = np.int32(40)
        name = "maze3d_disc_%s_s%d_%dx%d.npz" % (task_type.lower(), seed, res[0], res[1])
        np.savez_compressed(os.path.join(OUT, name), **d)
        print("wrote", name, "dones", int(d["done"].sum()), "obs max", int(d["obs"].max()))

    # ---- non-default renderer parameters ---------------------------------------------------------
    # MazeCoreDiscrete3D takes max_vision_range / fol_angle (maze_discrete_3d.py:18-37); the gym wrapper only
    # forwards the resolution, so the core of the registered env is swapped for one built with other
    # values (unmodified reference classes). 11x11 maze with 1.5-wide cells (not a power of two), dense food.
    from metagym.metamaze.envs.maze_discrete_3d import MazeCoreDiscrete3D
    for task_type, seed, res, vision, fov in (("SURVIVAL", 7, (90, 50), 6.0, 0.80), ("ESCAPE", 8, (37, 150), 20.0, 0.40)):
        task = _sample_task(seed, n=11, allow_loops=True, crowd_ratio=0.3, cell_size=1.5, wall_height=2.7,
                            agent_height=1.0, step_reward=-0.01, goal_reward=1.0, food_density=0.15, food_interval=4)
        env = gym.make("meta-maze-discrete-3D-v0", max_steps=40, enable_render=False, task_type=task_type,
                       resolution=res)
        env.maze_core = MazeCoreDiscrete3D(max_vision_range=vision, fol_angle=fov * 3.1415926, resolution_horizon=res[0],
                                           resolution_vertical=res[1], max_steps=40, task_type=task_type)
        actions = np.random.RandomState(seed + 1).choice(4, size=40, p=[0.25, 0.25, 0.1, 0.4])
        d = _run(env, task, actions, "disc", 2)
        d["resolution"] = np.asarray(res, np.int32)
        d["max_steps"] = np.int32(40)
        d["max_vision"] = np.float64(vision)
        d["fol_angle"] = np.float64(fov * 3.1415926)
        name = "maze3d_disc_%s_s%d_%dx%d.npz" % (task_type.lower(), seed, res[0], res[1])
        np.savez_compressed(os.path.join(OUT, name), **d)
        print("wrote", name, "dones", int(d["done"].sum()), "obs max", int(d["obs"].max()))

    # ---- MetaMazeContinuous3D ---------------------------------------------------------------------
    for task_type, seed in (("ESCAPE", 5), ("SURVIVAL", 6)):
        task = _sample_task(seed, n=9, allow_loops=False, step_reward=-0.001, goal_reward=1.0,
                            food_density=0.06, food_interval=5)
        env = gym.make("meta-maze-continuous-3D-v0", max_steps=50, enable_render=False, task_type=task_type,
                       resolution=(64, 64))
        rs = np.random.RandomState(seed + 1)
        actions = np.stack([rs.uniform(-1.2, 1.2, 70), rs.uniform(-0.6, 1.2, 70)], 1).astype(np.float32)
        d = _run(env, task, actions, "cont", 1)
        d["resolution"] = np.asarray((64, 64), np.int32)
        d["max_steps"] = np.int32(50)
        name = "maze3d_cont_%s_s%d.npz" % (task_type.lower(), seed)
        np.savez_compressed(os.path.join(OUT, name), **d)
        print("wrote", name, "dones", int(d["done"].sum()), "obs max", int(d["obs"].max()))
