"""Parity of the HIP MetaMaze kernels (through the C ABI / metagym_amd.metamaze) against
  (a) the golden vectors recorded from the unmodified reference (tests/golden/maze*.npz): transitions,
      reward (f64), done, steps, life and EVERY pixel of the recorded frames bit-exact for the
      2-D and discrete 3-D mazes; continuous 3-D within 1e-5 / >= 99.9 % identical pixels, and
  (b) the CPU oracle (oracle/maze_oracle.c) on batches of random tasks and actions.
GPU box only (-m gpu)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import maze as mo

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _files(pattern):
    return sorted(glob.glob(os.path.join(GOLDEN, pattern)))


def _tt(path):
    return "SURVIVAL" if "survival" in os.path.basename(path) else "ESCAPE"


@pytest.fixture(scope="module", autouse=True)
def reference_textures():
    """Render with the reference's textures (committed as uint8 in maze_textures.npz)."""
    from metagym_amd.metamaze import MAZE_TASK_MANAGER
    tex = np.load(os.path.join(GOLDEN, "maze_textures.npz"))
    MAZE_TASK_MANAGER.set_textures(tex["grounds"], tex["ceil"])
    yield


def _task_from_golden(g):
    from metagym_amd.metamaze import TaskConfig
    return TaskConfig(**{k[5:]: (tuple(int(x) for x in g[k]) if k[5:] in ("start", "goal") else
                                 (g[k] if g[k].ndim else g[k].item())) for k in g.files if k.startswith("task_")})


def _replay(env, g, continuous=False):
    """Drive a 1-env batch with the golden action stream, yield per-step outputs."""
    env.set_task(_task_from_golden(g))
    obs0 = env.reset().cpu().numpy()[0]
    out = []
    for t, a in enumerate(g["actions"]):
        if g["reset_before"][t]:
            env.reset()
        act = torch.as_tensor(np.asarray(a)[None], dtype=torch.float32 if continuous else torch.int32)
        obs, rew, done, info = env.step(act)
        out.append(dict(obs=obs, reward=float(env.reward64[0]), done=bool(done[0]), steps=int(info["steps"][0]),
                        grid=env.grid[:, 0].cpu().numpy(), life=float(env.life[0]), ori_idx=int(env.ori_idx[0]),
                        ori=float(env.ori[0]), loc=env.loc[:, 0].cpu().numpy()))
        out[-1]["obs_np"] = obs[0].cpu().numpy() if t in set(g["obs_step"].tolist()) else None
    return obs0, out


@pytest.mark.parametrize("path", _files("maze2d_*.npz"))
def test_maze2d_matches_reference_bit_exact(path):
    import metagym_amd
    g = np.load(path)
    env = metagym_amd.make("meta-maze-2D-v0", num_envs=1, device="cuda:0", max_steps=int(g["max_steps"]),
                           view_grid=int(g["view_grid"]), task_type=_tt(path))
    obs0, out = _replay(env, g)
    assert np.array_equal(obs0, g["obs0"])
    oi = 0
    for t, o in enumerate(out):
        assert list(o["grid"]) == list(g["grid"][t]), t
        assert o["reward"] == g["reward"][t] and o["done"] == bool(g["done"][t]) and o["steps"] == g["steps"][t], t
        if _tt(path) == "SURVIVAL":
            assert o["life"] == g["life"][t], t
        if o["obs_np"] is not None:
            assert np.array_equal(o["obs_np"], g["obs"][oi]), t
            oi += 1
    assert oi == len(g["obs_step"])


@pytest.mark.parametrize("path", _files("maze3d_disc_*.npz"))
def test_maze3d_discrete_matches_reference_pixel_exact(path):
    import metagym_amd
    g = np.load(path)
    res = tuple(int(x) for x in g["resolution"])
    env = metagym_amd.make("meta-maze-discrete-3D-v0", num_envs=1, device="cuda:0", max_steps=int(g["max_steps"]),
                           resolution=res, task_type=_tt(path))
    if "max_vision" in g.files:      # goldens recorded with non-default renderer parameters
        env.max_vision_range, env.fol_angle = float(g["max_vision"]), float(g["fol_angle"])
    obs0, out = _replay(env, g)
    assert obs0.dtype == np.int32 and obs0.shape == res + (3,)
    assert np.array_equal(obs0, g["obs0"]), "reset frame: %d values differ" % int((obs0 != g["obs0"]).sum())
    oi = 0
    for t, o in enumerate(out):
        assert list(o["grid"]) == list(g["grid"][t]) and o["ori_idx"] == g["ori_idx"][t], t
        assert o["reward"] == g["reward"][t] and o["done"] == bool(g["done"][t]) and o["steps"] == g["steps"][t], t
        if _tt(path) == "SURVIVAL":
            assert o["life"] == g["life"][t], t
        if o["obs_np"] is not None:
            bad = int((o["obs_np"] != g["obs"][oi]).sum())
            assert bad == 0, "step %d: %d differing values, max |d| %d" % (t, bad, int(np.abs(o["obs_np"] - g["obs"][oi]).max()))
            oi += 1
    assert oi == len(g["obs_step"])


@pytest.mark.parametrize("path", _files("maze3d_cont_*.npz"))
def test_maze3d_continuous_matches_reference(path):
    import metagym_amd
    g = np.load(path)
    res = tuple(int(x) for x in g["resolution"])
    env = metagym_amd.make("meta-maze-continuous-3D-v0", num_envs=1, device="cuda:0", max_steps=int(g["max_steps"]),
                           resolution=res, task_type=_tt(path))
    obs0, out = _replay(env, g, continuous=True)
    assert np.array_equal(obs0, g["obs0"])
    oi, bad, total, worst = 0, 0, 0, 0
    for t, o in enumerate(out):
        assert list(o["grid"]) == list(g["grid"][t]), t
        assert o["reward"] == g["reward"][t] and o["done"] == bool(g["done"][t]), t
        assert np.allclose(o["loc"], g["loc"][t], rtol=1e-5, atol=1e-5), t        # north-star 1e-5
        assert abs(o["ori"] - g["ori"][t]) <= 1e-5 * max(1.0, abs(g["ori"][t])), t
        if o["obs_np"] is not None:
            diff = o["obs_np"] != g["obs"][oi]
            bad += int(diff.sum())
            total += diff.size
            if diff.any():
                worst = max(worst, int(np.abs(o["obs_np"] - g["obs"][oi]).max()))
            oi += 1
    print(os.path.basename(path), "mismatching values %d / %d, max |d| %d" % (bad, total, worst))
    assert bad <= 1e-3 * total


# ---- batches against the oracle -------------------------------------------------------------------

def _oracle_batch(tasks, task_ids, tt):
    otasks = [mo.Task(**t._asdict()) for t in tasks]
    states = [mo.State(otasks[i]) for i in task_ids]
    for s, i in zip(states, task_ids):
        mo.reset(otasks[i], tt, s)
    return otasks, states


def test_maze2d_batch_matches_oracle():
    """BASELINE config C1 scaled out: 15x15 mazes, 8 tasks, 3000 envs (ragged), 80 steps with
    masked resets of finished envs; everything bit-exact."""
    import metagym_amd
    from metagym_amd.metamaze import MazeTaskSampler
    for task_type in ("ESCAPE", "SURVIVAL"):
        tt = mo.TASK_TYPES[task_type]
        tasks = [MazeTaskSampler(n=15, allow_loops=True, crowd_ratio=0.35, step_reward=-0.01, goal_reward=1.0,
                                 food_density=0.08, food_interval=6, seed=s) for s in range(8)]
        n = 3000 + 7
        env = metagym_amd.make("meta-maze-2D-v0", num_envs=n, device="cuda:0", max_steps=25, view_grid=1,
                               task_type=task_type)
        env.set_task(tasks)
        ids = env.task_id.cpu().numpy()
        otasks, states = _oracle_batch(tasks, ids, tt)
        obs = env.reset().cpu().numpy()
        for e in range(0, n, 97):
            assert np.array_equal(obs[e], mo.observe_2d(otasks[ids[e]], tt, states[e], 1))
        rs = np.random.RandomState(5)
        for t in range(80):
            a = rs.randint(0, 4, n)
            obs, rew, done, info = env.step(torch.as_tensor(a))
            r64, d, ob = env.reward64.cpu().numpy(), done.cpu().numpy(), obs.cpu().numpy()
            for e in range(n):
                r, dd = mo.step_2d(otasks[ids[e]], tt, 25, states[e], a[e])
                assert r == r64[e] and dd == d[e], (t, e)
            for e in range(0, n, 61):
                assert np.array_equal(ob[e], mo.observe_2d(otasks[ids[e]], tt, states[e], 1)), (t, e)
            if d.any():
                env.reset(mask=done)
                for e in np.nonzero(d)[0]:
                    mo.reset(otasks[ids[e]], tt, states[e])
        assert np.array_equal(env.grid.cpu().numpy().T, np.asarray([list(s.c.grid) for s in states]))
        if task_type == "SURVIVAL":
            # the kernel keeps the SURVIVAL arrays per food SLOT ([max_food, N], mg_maze_state.food_by_slot); state_dict hands
            # them out by CELL like the reference's arrays — every cell of every env against the oracle — and reads them back
            assert env.cur_food.shape == (env._max_food, n) and env._max_food < 15 * 15
            sd = env.state_dict()
            for key, get in (("cur_food", lambda s: s.cur_food), ("wait_refresh", lambda s: s.wait), ("revival", lambda s: s.revival)):
                got = sd[key].cpu().numpy()
                assert got.shape == (15 * 15, n)
                assert np.array_equal(got.T, np.asarray([get(s) for s in states])), key
            twin = metagym_amd.make("meta-maze-2D-v0", num_envs=n, device="cuda:0", max_steps=25, view_grid=1, task_type=task_type)
            twin.set_task(tasks, task_ids=torch.zeros(n, dtype=torch.int32))      # (other tasks: load_state_dict brings task_id along)
            twin.reset()
            twin.load_state_dict(sd)
            for t in range(6):
                a = torch.as_tensor(rs.randint(0, 4, n))
                ra, rb = env.step(a), twin.step(a)
                assert torch.equal(ra[0], rb[0]) and torch.equal(env.reward64, twin.reward64) and torch.equal(ra[2], rb[2])
                env.reset(mask=ra[2]); twin.reset(mask=rb[2])


def test_maze2d_survival_crumbs_outside_the_food_list():
    """Cells the per-slot layout does not store: a task value of exactly 0.0 is never read (cell_slot -1), a nonzero value
    <= 1e-2 — never eaten (maze_base.py:71 needs > 1e-2), never renewed (interval 0) — still shows in the 2-D observation
    (maze_2d.py:117) and is read from the task table (cell_slot -2). Hand-edited tasks with such crumbs on half of the free
    cells, 40 steps, everything bit-exact against the oracle; and a checkpoint whose counters are negative where nothing waits
    (unreachable by reset / step; the slot kernel would not see them) is refused."""
    import metagym_amd
    from metagym_amd.metamaze import MazeTaskSampler
    tt = mo.TASK_TYPES["SURVIVAL"]
    rs = np.random.RandomState(9)
    tasks = []
    for s_ in range(4):
        t = MazeTaskSampler(n=9, allow_loops=True, step_reward=-0.01, goal_reward=1.0, food_density=0.08, food_interval=5, seed=s_)
        food = np.array(t.food_rewards, np.float64)
        free = (np.array(t.cell_walls) == 0) & (food == 0.0)
        crumbs = free & (rs.rand(*food.shape) < 0.5)
        food[crumbs] = rs.uniform(1e-4, 1e-2, int(crumbs.sum()))
        tasks.append(t._replace(food_rewards=food))
    # a fifth task with MANY food cells: the other tasks' rows of the padded food list (max_food wide) then run on into their
    # non-food cells — crumbs among them — and the cell -> slot table must leave those at -2 (a padding entry used
    # to be scattered as -1, which hid the crumb from the observation)
    t = MazeTaskSampler(n=9, allow_loops=True, step_reward=-0.01, goal_reward=1.0, food_density=0.5, food_interval=5, seed=11)
    tasks.append(t)
    n = 257
    env = metagym_amd.make("meta-maze-2D-v0", num_envs=n, device="cuda:0", max_steps=30, view_grid=1, task_type="SURVIVAL")
    env.set_task(tasks)
    slots = env._cell_slot_t.cpu().numpy()
    assert (slots == -2).sum() > 20 and (slots == -1).sum() > 20 and (slots >= 0).sum() > 4
    n_food, cells = env._n_food_t.cpu().numpy(), env._food_cells_t.cpu().numpy()
    assert n_food.max() - n_food.min() >= 12
    padded_crumbs = 0
    for ti, tk in enumerate(tasks):
        food = np.asarray(tk.food_rewards, np.float64).reshape(-1)
        for c in cells[ti, n_food[ti]:]:
            if 0.0 < food[c] <= 1e-2:
                padded_crumbs += 1
                assert slots[ti, c] == -2, (ti, c)
        for c in range(81):
            want = -2 if (0.0 < food[c] <= 1e-2) else (-1 if food[c] == 0.0 and slots[ti, c] < 0 else slots[ti, c])
            assert slots[ti, c] == want, (ti, c)
    assert padded_crumbs > 0                                  # the regression is exercised
    ids = env.task_id.cpu().numpy()
    otasks, states = _oracle_batch(tasks, ids, tt)
    obs = env.reset().cpu().numpy()
    seen_crumb = 0
    for t in range(40):
        a = rs.randint(0, 4, n)
        obs, rew, done, info = env.step(torch.as_tensor(a))
        r64, d, ob = env.reward64.cpu().numpy(), done.cpu().numpy(), obs.cpu().numpy()
        for e in range(n):
            r, dd = mo.step_2d(otasks[ids[e]], tt, 30, states[e], a[e])
            assert r == r64[e] and dd == d[e], (t, e)
            want = mo.observe_2d(otasks[ids[e]], tt, states[e], 1)
            assert np.array_equal(ob[e], want), (t, e)
            seen_crumb += int(((want > 0) & (want <= 1e-2)).any())
        if d.any():
            env.reset(mask=done)
            for e in np.nonzero(d)[0]:
                mo.reset(otasks[ids[e]], tt, states[e])
    assert seen_crumb > n                                      # crumbs were in view all along
    sd = env.state_dict()
    bad = {k: v.clone() for k, v in sd.items()}
    cell = int(np.argmax(slots[ids[0]] >= 0))                   # a listed cell of env 0's task
    bad["wait_refresh"][cell, 0], bad["revival"][cell, 0] = 0, -3
    with pytest.raises(ValueError, match="not waiting"):
        env.load_state_dict(bad)
    env.load_state_dict(sd)


@pytest.mark.parametrize("continuous", [False, True])
def test_maze3d_batch_matches_oracle(continuous):
    """9x9 mazes (config C3 geometry), 6 tasks x 96 envs, 64x64 and 40x24 frames, both task types;
    discrete frames must be identical to the oracle's, continuous >= 99.9 %."""
    import metagym_amd
    from metagym_amd.metamaze import MazeTaskSampler, MAZE_TASK_MANAGER
    tex_u8 = MAZE_TASK_MANAGER.grounds.astype(np.uint8)
    for task_type, res in (("SURVIVAL", (64, 64)), ("ESCAPE", (40, 24))):
        tt = mo.TASK_TYPES[task_type]
        tasks = [MazeTaskSampler(n=9, allow_loops=False, step_reward=-0.01, goal_reward=1.0, food_density=0.08,
                                 food_interval=4, seed=10 + s) for s in range(6)]
        n = 96
        name = "meta-maze-continuous-3D-v0" if continuous else "meta-maze-discrete-3D-v0"
        env = metagym_amd.make(name, num_envs=n, device="cuda:0", max_steps=12, resolution=res, task_type=task_type)
        env.set_task(tasks)
        ids = env.task_id.cpu().numpy()
        otasks, states = _oracle_batch(tasks, ids, tt)
        view = mo.View(tex_u8, MAZE_TASK_MANAGER.ceil, res[0], res[1])
        obs = env.reset().cpu().numpy()
        for e in range(0, n, 13):
            assert np.array_equal(obs[e], mo.observe_3d(otasks[ids[e]], tt, view, states[e], int(continuous)))
        rs = np.random.RandomState(9)
        bad = total = 0
        for t in range(20):
            if continuous:
                a = np.stack([rs.uniform(-1.2, 1.2, n), rs.uniform(-0.5, 1.2, n)], 1).astype(np.float32)
            else:
                a = rs.choice(4, size=n, p=[0.2, 0.2, 0.1, 0.5])
            obs, rew, done, info = env.step(torch.as_tensor(a))
            r64, d, ob = env.reward64.cpu().numpy(), done.cpu().numpy(), obs.cpu().numpy()
            for e in range(n):
                if continuous:
                    r, dd = mo.step_cont3d(otasks[ids[e]], tt, 12, states[e], a[e][0], a[e][1])
                else:
                    r, dd = mo.step_disc3d(otasks[ids[e]], tt, 12, states[e], a[e])
                assert r == r64[e] and dd == d[e], (t, e)
            for e in range(t % 7, n, 7):
                ref = mo.observe_3d(otasks[ids[e]], tt, view, states[e], int(continuous))
                bad += int((ob[e] != ref).sum())
                total += ref.size
            if d.any():
                env.reset(mask=done)
                for e in np.nonzero(d)[0]:
                    mo.reset(otasks[ids[e]], tt, states[e])
        print(task_type, res, "continuous" if continuous else "discrete", "pixel mismatches", bad, "/", total)
        assert bad == 0 if not continuous else bad <= 1e-3 * total


@pytest.mark.parametrize("continuous", [False, True], ids=["discrete", "continuous"])
@pytest.mark.parametrize("res", [(64, 64), (72, 40), (128, 32), (24, 24), (40, 24), (64, 48), (48, 20), (60, 60)],
                         ids=lambda r: "%dx%d" % r)
def test_maze3d_small_frame_renderer_every_pixel(res, continuous):
    """The one-wave-per-env instantiation of the renderer (maze3d_step_kernel<., ., SMALL>, frames up to 64 x 64 pixels: the column
    record broadcast through the LDS crossbar, the frame store deferred by one chunk and issued as a buffer store whose range is the
    env's frame; ray_caster_utils.py:66-209): every pixel of every frame against the oracle for frame shapes that exercise its
    corners — a full 64-row chunk per column, ragged last chunks (40, 24, 20, 60 rows: lanes past V must store nothing, and with a
    deferred store "nothing" is an out-of-range buffer offset), several column groups per wave (72 = 32 + 32 + 8, 128), both task
    types (life bar; translucent food cells / the goal overlay), food eaten and re-grown inside the compared steps; discrete
    frames bit-identical, continuous ones to the usual 99.9 %. (Round 5 also built a transposed renderer on these shapes — lane =
    column in the pixel pass too — which passed this test and was 2.7x slower: profiles/r05/maze3d_small_frames.txt.)"""
    import metagym_amd
    from metagym_amd.metamaze import MazeTaskSampler, MAZE_TASK_MANAGER
    tex_u8 = MAZE_TASK_MANAGER.grounds.astype(np.uint8)
    name = "meta-maze-continuous-3D-v0" if continuous else "meta-maze-discrete-3D-v0"
    n = 48
    for task_type in ("SURVIVAL", "ESCAPE"):
        tt = mo.TASK_TYPES[task_type]
        tasks = [MazeTaskSampler(n=9, allow_loops=False, step_reward=-0.01, goal_reward=1.0, food_density=0.15,
                                 food_interval=3, seed=70 + s_) for s_ in range(6)]
        env = metagym_amd.make(name, num_envs=n, device="cuda:0", max_steps=40, resolution=res, task_type=task_type)
        env.set_task(tasks)
        ids = env.task_id.cpu().numpy()
        otasks, states = _oracle_batch(tasks, ids, tt)
        view = mo.View(tex_u8, MAZE_TASK_MANAGER.ceil, res[0], res[1])
        obs = env.reset().cpu().numpy()
        rs = np.random.RandomState(res[0] * 100 + res[1])
        bad = total = 0
        for t in range(10):
            for e in range(n):
                want = mo.observe_3d(otasks[ids[e]], tt, view, states[e], int(continuous))
                bad += int((obs[e] != want).sum())
                total += want.size
            if continuous:
                a = np.stack([rs.uniform(-1.2, 1.2, n), rs.uniform(-0.5, 1.2, n)], 1).astype(np.float32)
            else:
                a = rs.choice(4, size=n, p=[0.2, 0.2, 0.1, 0.5])
            obs, rew, done, info = env.step(torch.as_tensor(a))
            obs, r64, d = obs.cpu().numpy(), env.reward64.cpu().numpy(), done.cpu().numpy()
            for e in range(n):
                if continuous:
                    r, dd = mo.step_cont3d(otasks[ids[e]], tt, 40, states[e], float(a[e, 0]), float(a[e, 1]))
                else:
                    r, dd = mo.step_disc3d(otasks[ids[e]], tt, 40, states[e], int(a[e]))
                assert r == r64[e] and dd == d[e], (task_type, t, e)
            if d.any():
                obs = env.reset(mask=done).cpu().numpy()
                for e in np.nonzero(d)[0]:
                    mo.reset(otasks[ids[e]], tt, states[e])
        if continuous:
            assert bad <= 1e-3 * total, (task_type, bad, total)
        else:
            assert bad == 0, (task_type, bad, total)


def test_maze3d_ragged_batch_every_env_every_frame():
    """29 envs = three full groups of 8 (whose env -> workgroup assignment is rotated, mg::env_of_block) plus a
    ragged group of 5 (identity): every env's reward, done and full frame against the oracle on every step."""
    import metagym_amd
    from metagym_amd.metamaze import MazeTaskSampler, MAZE_TASK_MANAGER
    tex_u8 = MAZE_TASK_MANAGER.grounds.astype(np.uint8)
    tt = mo.TASK_TYPES["SURVIVAL"]
    tasks = [MazeTaskSampler(n=9, allow_loops=False, step_reward=-0.01, goal_reward=1.0, food_density=0.08,
                             food_interval=4, seed=40 + s) for s in range(5)]
    n, res = 29, (32, 24)
    env = metagym_amd.make("meta-maze-discrete-3D-v0", num_envs=n, device="cuda:0", max_steps=50, resolution=res,
                           task_type="SURVIVAL")
    env.set_task(tasks)
    ids = env.task_id.cpu().numpy()
    otasks, states = _oracle_batch(tasks, ids, tt)
    view = mo.View(tex_u8, MAZE_TASK_MANAGER.ceil, res[0], res[1])
    env.reset()
    rs = np.random.RandomState(3)
    for t in range(6):
        a = rs.choice(4, size=n, p=[0.2, 0.2, 0.1, 0.5])
        obs, rew, done, info = env.step(torch.as_tensor(a))
        r64, d, ob = env.reward64.cpu().numpy(), done.cpu().numpy(), obs.cpu().numpy()
        for e in range(n):
            r, dd = mo.step_disc3d(otasks[ids[e]], tt, 50, states[e], a[e])
            assert r == r64[e] and dd == d[e], (t, e)
            assert np.array_equal(ob[e], mo.observe_3d(otasks[ids[e]], tt, view, states[e], 0)), (t, e)


def test_maze3d_full_size_properties():
    """Config C3 size (16 384 envs, 9x9) at 32x32 frames: determinism, env-permutation equivariance
    and auto-reset == explicit masked reset; plus one 256x256 batch checked against the oracle."""
    import metagym_amd
    from metagym_amd.metamaze import MazeTaskSampler, MAZE_TASK_MANAGER
    tasks = [MazeTaskSampler(n=9, allow_loops=False, step_reward=-0.01, goal_reward=1.0, food_density=0.08,
                             food_interval=4, seed=100 + s) for s in range(64)]
    n = 16384
    mk = lambda **kw: metagym_amd.make("meta-maze-discrete-3D-v0", num_envs=n, device="cuda:0", max_steps=6,
                                       resolution=(32, 32), task_type="SURVIVAL", **kw)
    a, b, c = mk(), mk(), mk(auto_reset=True)
    ids = torch.arange(n, dtype=torch.int32) % 64
    perm = torch.randperm(n)
    a.set_task(tasks, ids)
    b.set_task(tasks, ids[perm])
    c.set_task(tasks, ids)
    oa, ob, oc = a.reset(), b.reset(), c.reset()
    assert torch.equal(oa[perm.cuda()], ob) and torch.equal(oa, oc)
    g = torch.Generator().manual_seed(0)
    for t in range(14):
        act = torch.randint(0, 4, (n,), generator=g, dtype=torch.int32)
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act[perm])
        oc, rc, dc, _ = c.step(act)
        assert torch.equal(oa[perm.cuda()], ob) and torch.equal(ra[perm.cuda()], rb) and torch.equal(da[perm.cuda()], db)
        assert torch.equal(ra, rc) and torch.equal(da, dc)
        if bool(da.any()):
            oa = a.reset(mask=da)
            b.reset(mask=db)
        assert torch.equal(oa, oc), "auto-reset differs from explicit reset at step %d" % t
    # 256x256 (the registered default resolution), 32 envs, against the oracle
    tt = mo.SURVIVAL
    env = metagym_amd.make("meta-maze-discrete-3D-v0", num_envs=32, device="cuda:0", max_steps=50, task_type="SURVIVAL")
    env.set_task(tasks[:8])
    idl = env.task_id.cpu().numpy()
    otasks, states = _oracle_batch(tasks[:8], idl, tt)
    view = mo.View(MAZE_TASK_MANAGER.grounds.astype(np.uint8), MAZE_TASK_MANAGER.ceil, 256, 256)
    env.reset()
    rs = np.random.RandomState(1)
    for t in range(5):
        act = rs.choice(4, size=32, p=[0.2, 0.2, 0.1, 0.5])
        obs, _, _, _ = env.step(torch.as_tensor(act))
        for e in range(32):
            mo.step_disc3d(otasks[idl[e]], tt, 50, states[e], act[e])
    ob = obs.cpu().numpy()
    for e in (0, 7, 19, 31):
        assert np.array_equal(ob[e], mo.observe_3d(otasks[idl[e]], tt, view, states[e], 0)), e


def _c3_sample_envs(n):
    """64 envs of an n-env batch: the first and the last workgroup group of 8, and one whole aligned group of 8 for every
    value of the env -> XCD rotation mg::env_of_block applies (metagym_amd/csrc/mg_common.h), spread over the launch."""
    rot = lambda q: (q + (q >> 3) + (q >> 6) + (q >> 9)) & 7
    groups, seen = [0, n // 8 - 1], {rot(0), rot(n // 8 - 1)}
    q = 37
    while len(groups) < 8:
        if rot(q) not in seen or len(seen) == 8:
            groups.append(q)
            seen.add(rot(q))
        q = (q + 263) % (n // 8)
    assert len({rot(g) for g in groups}) >= 6
    return np.asarray(sorted(8 * g + x for g in groups for x in range(8)))


@pytest.mark.parametrize("continuous", [False, True], ids=["discrete", "continuous"])
def test_c3_timed_configuration_sampled_against_oracle(continuous):
    """BASELINE configs[2] exactly as bench.py times it (secondary_workloads: 16 384 envs, 9x9 mazes, 64 tasks of
    bench.maze_tasks(), SURVIVAL, 256x256 frames, fused auto-reset), with max_steps short enough that every env ends
    episodes inside the 10 compared steps. 64 sampled envs are replayed on oracle/maze_oracle.c: reward, done, the whole
    per-env state and EVERY pixel of every sampled frame. Discrete: bit-exact. Continuous: loc / heading within the
    north-star 1e-5 (bit-equal in practice) and every pixel exact wherever the pose is bit-equal."""
    import metagym_amd
    from metagym_amd.metamaze import MazeTaskSampler, MAZE_TASK_MANAGER
    n, res, max_steps, n_steps = 16384, 256, 4, 10
    tasks = [MazeTaskSampler(n=9, allow_loops=False, step_reward=-0.01, goal_reward=1.0, food_density=0.06,
                             food_interval=20, seed=s) for s in range(64)]            # == bench.maze_tasks()
    ident = "meta-maze-continuous-3D-v0" if continuous else "meta-maze-discrete-3D-v0"
    env = metagym_amd.make(ident, num_envs=n, device="cuda:0", max_steps=max_steps, resolution=(res, res),
                           task_type="SURVIVAL", auto_reset=True)
    env.set_task(tasks)
    ids = env.task_id.cpu().numpy()
    assert np.array_equal(ids, np.arange(n) % 64)
    sample = _c3_sample_envs(n)
    sample_t = torch.as_tensor(sample).cuda()
    tt = mo.SURVIVAL
    otasks = [mo.Task(**t._asdict()) for t in tasks]
    states = {int(e): mo.State(otasks[ids[e]]) for e in sample}
    for e in sample:
        mo.reset(otasks[ids[e]], tt, states[int(e)])
    view = mo.View(MAZE_TASK_MANAGER.grounds.astype(np.uint8), MAZE_TASK_MANAGER.ceil, res, res)

    def frames_match(obs, t):
        ob = obs[sample_t].cpu().numpy()
        bad = total = 0
        for k, e in enumerate(sample):
            s = states[int(e)]
            ref = mo.observe_3d(otasks[ids[e]], tt, view, s, int(continuous))
            d = int((ob[k] != ref).sum())
            if continuous:
                loc, ori = env.loc[:, int(e)].cpu().numpy(), float(env.ori[int(e)])
                assert np.allclose(loc, np.asarray(s.c.loc[:]), rtol=1e-5, atol=1e-5), (t, e)
                assert abs(ori - s.c.ori) <= 1e-5 * max(1.0, abs(s.c.ori)), (t, e)
                if np.array_equal(loc, np.asarray(s.c.loc[:], np.float32)) and ori == s.c.ori:
                    assert d == 0, "step %d env %d: pose bit-equal but %d pixel values differ" % (t, e, d)
            else:
                assert d == 0, "step %d env %d: %d pixel values differ" % (t, e, d)
            bad += d
            total += ref.size
        return bad, total

    bad, total = frames_match(env.reset(), -1)
    rs = np.random.RandomState(11 + int(continuous))
    ends = 0
    for t in range(n_steps):
        if continuous:
            a = np.stack([rs.uniform(-1.2, 1.2, n), rs.uniform(-0.5, 1.2, n)], 1).astype(np.float32)
        else:
            a = rs.choice(4, size=n, p=[0.2, 0.2, 0.1, 0.5]).astype(np.int32)
        obs, rew, done, info = env.step(torch.as_tensor(a))
        r64, d = env.reward64[sample_t].cpu().numpy(), done[sample_t].cpu().numpy()
        for k, e in enumerate(sample):
            s, task = states[int(e)], otasks[ids[e]]
            if continuous:
                r, dd = mo.step_cont3d(task, tt, max_steps, s, a[e][0], a[e][1])
            else:
                r, dd = mo.step_disc3d(task, tt, max_steps, s, a[e])
            assert r == r64[k] and dd == bool(d[k]), (t, e)
            if dd:                                          # fused auto-reset: the env restarts inside the launch
                mo.reset(task, tt, s)
                ends += 1
        b, tot = frames_match(obs, t)
        bad, total = bad + b, total + tot
        # the whole per-env state of the sampled envs
        grid, steps, life = env.grid[:, sample_t].cpu().numpy(), env.steps[sample_t].cpu().numpy(), env.life[sample_t].cpu().numpy()
        food, wait, rev = (x[sample_t].cpu().numpy() for x in (env.cur_food, env.wait_refresh, env.revival))
        oidx = env.ori_idx[sample_t].cpu().numpy()
        for k, e in enumerate(sample):
            s = states[int(e)]
            assert (grid[0, k], grid[1, k], steps[k]) == (s.c.grid[0], s.c.grid[1], s.c.steps), (t, e)
            assert life[k] == s.c.life, (t, e)
            assert np.array_equal(food[k], s.cur_food) and np.array_equal(wait[k], s.wait) and np.array_equal(rev[k], s.revival), (t, e)
            if not continuous:
                assert oidx[k] == s.c.ori_idx, (t, e)
    assert ends >= 2 * len(sample), "episode ends inside the compared steps: %d" % ends
    print("C3 %s: %d sampled envs x %d frames, %d episode ends, pixel mismatches %d / %d"
          % ("continuous" if continuous else "discrete", len(sample), n_steps + 1, ends, bad, total))
    assert bad <= (1e-3 * total if continuous else 0)


@pytest.mark.parametrize("n,res,cell", [(15, (64, 48), 2.0), (21, (32, 32), 2.0), (15, (32, 32), 0.75),
                                        # ragged frame on the 4-waves-per-env path: 136 columns = 4 slabs of 32 + 8,
                                        # 150 rows = 64 + 64 + 22; cell size not a power of two (true divisions)
                                        (15, (136, 150), 1.5), (9, (70, 200), 2.0)])
def test_maze3d_larger_mazes_match_oracle(n, res, cell):
    """The reference's default maze size is 15x15 (maze_task.py:42); larger grids and a small cell size
    (many cells inside the vision range, so many overlay records per ray and > 64 KiB of LDS) must
    render identically to the oracle too. Dense food so most rays cross translucent cells."""
    import metagym_amd
    from metagym_amd.metamaze import MazeTaskSampler, MAZE_TASK_MANAGER
    tt = mo.SURVIVAL
    tasks = [MazeTaskSampler(n=n, allow_loops=True, crowd_ratio=0.25, cell_size=cell, wall_height=1.6 * cell,
                             agent_height=0.8 * cell, step_reward=-0.01, goal_reward=1.0, food_density=0.3,
                             food_interval=3, seed=40 + s) for s in range(3)]
    env = metagym_amd.make("meta-maze-discrete-3D-v0", num_envs=12, device="cuda:0", max_steps=30, resolution=res,
                           task_type="SURVIVAL")
    env.set_task(tasks)
    ids = env.task_id.cpu().numpy()
    otasks, states = _oracle_batch(tasks, ids, tt)
    view = mo.View(MAZE_TASK_MANAGER.grounds.astype(np.uint8), MAZE_TASK_MANAGER.ceil, res[0], res[1])
    obs = env.reset().cpu().numpy()
    assert np.array_equal(obs[0], mo.observe_3d(otasks[ids[0]], tt, view, states[0], 0))
    rs = np.random.RandomState(n)
    for t in range(10):
        a = rs.choice(4, size=12, p=[0.2, 0.2, 0.1, 0.5])
        obs, _, done, _ = env.step(torch.as_tensor(a))
        ob = obs.cpu().numpy()
        for e in range(12):
            mo.step_disc3d(otasks[ids[e]], tt, 30, states[e], a[e])
        for e in range(t % 3, 12, 3):
            ref = mo.observe_3d(otasks[ids[e]], tt, view, states[e], 0)
            assert np.array_equal(ob[e], ref), (t, e, int((ob[e] != ref).sum()))


def test_wrong_uniform_cell_size_is_an_error_code_not_wrong_pixels():
    """mg_maze_view.uniform_cell_size through the C ABI: a task table whose cell sizes differ, stepped with a
    view that vouches for one of them, comes back as MG_ERR_BAD_CONFIG from mg_maze3d_step itself (the library checks the pair the
    first time it sees it); the right value — and 0, "unknown" — render the oracle's frames; a table rewritten in place is caught
    by the explicit re-check; and a captured step finds the pair already checked by set_task."""
    import metagym_amd
    from metagym_amd import _lib
    from metagym_amd.metamaze import MazeTaskSampler, MAZE_TASK_MANAGER
    tt = mo.TASK_TYPES["ESCAPE"]
    tasks = [MazeTaskSampler(n=9, allow_loops=False, step_reward=-0.01, goal_reward=1.0, seed=s_) for s_ in range(3)]
    env = metagym_amd.make("meta-maze-discrete-3D-v0", num_envs=6, device="cuda:0", max_steps=20, resolution=(32, 32), task_type="ESCAPE")
    env.set_task(tasks)
    cs = float(tasks[0].cell_size)
    assert env._uniform_cell_size == cs
    want = env.reset().clone()
    ids = env.task_id.cpu().numpy()
    otasks, states = _oracle_batch(tasks, ids, tt)
    view = mo.View(MAZE_TASK_MANAGER.grounds.astype(np.uint8), MAZE_TASK_MANAGER.ceil, 32, 32)
    for e in range(6):
        assert np.array_equal(want[e].cpu().numpy(), mo.observe_3d(otasks[ids[e]], tt, view, states[e], 0))
    # (1) a wrong promise: an error code from the step, frames untouched
    env._view_c.uniform_cell_size = cs * 0.5
    env._obs.fill_(-7)
    with pytest.raises(_lib.MetaGymHipError, match="uniform_cell_size"):
        env.step(torch.zeros(6, dtype=torch.int32))
    torch.cuda.synchronize()
    assert int((env._obs != -7).sum()) == 0
    # (2) "unknown" renders the same frames through the general kernel
    env._view_c.uniform_cell_size = 0.0
    assert torch.equal(env._observe(), want)
    env._view_c.uniform_cell_size = cs
    assert torch.equal(env._observe(), want)
    # (3) the table rewritten in place: one task's cell size changes under a checked pair -> the explicit re-check refuses it
    lib = env._lib
    sc = env._task_t["scalars"]
    keep = sc[1, 0].item()
    sc[1, 0] = keep * 2.0
    torch.cuda.synchronize()
    assert lib.mg_maze_check_uniform_cell_size(env._tasks_c, cs, None) == -1003 and b"task 1 of 3" in lib.mg_last_error()
    with pytest.raises(_lib.MetaGymHipError, match="task 1 of 3"):          # ... and the step no longer holds the pair as checked
        env.step(torch.zeros(6, dtype=torch.int32))
    sc[1, 0] = keep
    torch.cuda.synchronize()
    assert lib.mg_maze_check_uniform_cell_size(env._tasks_c, cs, None) == 0
    # (4) mixed cell sizes through set_task: the Python layer passes 0 and the frames still match the oracle
    mixed = [tasks[0], tasks[1]._replace(cell_size=cs * 0.75), tasks[2]]
    env2 = metagym_amd.make("meta-maze-discrete-3D-v0", num_envs=6, device="cuda:0", max_steps=20, resolution=(32, 32), task_type="ESCAPE")
    env2.set_task(mixed)
    assert env2._uniform_cell_size == 0.0
    f2 = env2.reset().cpu().numpy()
    ids2 = env2.task_id.cpu().numpy()
    ot2, st2 = _oracle_batch(mixed, ids2, tt)
    for e in range(6):
        assert np.array_equal(f2[e], mo.observe_3d(ot2[ids2[e]], tt, view, st2[e], 0))
    # (5) a step captured into a hipGraph right after set_task: the pair is already checked, capture does not synchronise
    env3 = metagym_amd.make("meta-maze-discrete-3D-v0", num_envs=6, device="cuda:0", max_steps=20, resolution=(32, 32), task_type="ESCAPE")
    env3.set_task(tasks)
    env3.reset()
    act = torch.ones(6, dtype=torch.int32, device="cuda:0")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        env3.step(act)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        env3.step(act)
    g.replay()
    torch.cuda.synchronize()


def test_maze3d_uint8_fast_path_is_the_clamped_reference_frame():
    """obs_dtype=torch.uint8 (non-parity fast path): every byte equals min(int32 frame, 255)."""
    import metagym_amd
    from metagym_amd.metamaze import MazeTaskSampler
    tasks = [MazeTaskSampler(n=9, allow_loops=False, food_density=0.08, food_interval=4, seed=70 + s) for s in range(4)]
    mk = lambda dt: metagym_amd.make("meta-maze-discrete-3D-v0", num_envs=40, device="cuda:0", max_steps=30,
                                     resolution=(48, 64), task_type="SURVIVAL", obs_dtype=dt)
    a, b = mk(torch.int32), mk(torch.uint8)
    a.set_task(tasks)
    b.set_task(tasks)
    oa, ob = a.reset(), b.reset()
    assert ob.dtype == torch.uint8 and torch.equal(oa.clamp(0, 255).to(torch.uint8), ob)
    g = torch.Generator().manual_seed(1)
    for _ in range(6):
        act = torch.randint(0, 4, (40,), generator=g, dtype=torch.int32)
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        assert torch.equal(oa.clamp(0, 255).to(torch.uint8), ob) and torch.equal(ra, rb) and torch.equal(da, db)
    assert int(oa.max()) > 255          # the int32 frames really do exceed a byte


@pytest.mark.parametrize("ident,res", [("meta-maze-discrete-3D-v0", (40, 100)), ("meta-maze-discrete-3D-v0", (24, 30)),
                                       ("meta-maze-discrete-3D-v0", (64, 64)), ("meta-maze-continuous-3D-v0", (32, 132)),
                                       ("meta-maze-discrete-3D-v0", (20, 256)), ("meta-maze-continuous-3D-v0", (33, 7))])
def test_maze3d_uint8_packed_store_shapes(ident, res):
    """The uint8 frames' packed store (round 6: a quad of lanes writes its 12 bytes as one dwordx3, csrc/maze.hip flush()) over
    the shapes that decide its path: res_v a multiple of 4 with a ragged last 64-row chunk (100, 132), one chunk exactly (64),
    four chunks (256), and heights that are not a multiple of 4 (30, 7: the byte-store path) — every byte equals
    min(int32 frame, 255), discrete and continuous, small-frame and four-wave kernels."""
    import metagym_amd
    from metagym_amd.metamaze import MazeTaskSampler
    tasks = [MazeTaskSampler(n=9, allow_loops=False, food_density=0.08, food_interval=4, seed=170 + s) for s in range(3)]
    n = 24
    mk = lambda dt: metagym_amd.make(ident, num_envs=n, device="cuda:0", max_steps=30, resolution=res, task_type="SURVIVAL", obs_dtype=dt)
    a, b = mk(torch.int32), mk(torch.uint8)
    a.set_task(tasks)
    b.set_task(tasks)
    oa, ob = a.reset(), b.reset()
    assert ob.dtype == torch.uint8 and ob.shape == oa.shape and torch.equal(oa.clamp(0, 255).to(torch.uint8), ob)
    g = torch.Generator().manual_seed(2)
    for _ in range(5):
        if "continuous" in ident:
            act = torch.rand(n, 2, generator=g) * 2 - 1
        else:
            act = torch.randint(0, 4, (n,), generator=g, dtype=torch.int32)
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        assert torch.equal(oa.clamp(0, 255).to(torch.uint8), ob) and torch.equal(ra, rb) and torch.equal(da, db)


def test_device_task_sampler_matches_reference_tasks_bit_exact(reference_textures):
    """mg_maze_sample_tasks (one wave per task, MT19937 streams on the device) against tasks drawn by the
    unmodified reference sampler: every field of every task, bit for bit (floats included)."""
    import json
    from metagym_amd.metamaze import MAZE_TASK_MANAGER
    g = np.load(os.path.join(GOLDEN, "maze_tasks.npz"))
    assert MAZE_TASK_MANAGER.n_texts == int(g["n_texts"])
    seeds = [int(s) for s in g["seeds"]]
    for c, kw in enumerate(json.loads(str(g["cases"]))):
        table = MAZE_TASK_MANAGER.sample_tasks_device(len(seeds), device="cuda:0", seeds=seeds, **kw)
        tasks = table.to_task_configs()
        for t, seed in zip(tasks, seeds):
            k = "c%d_s%d_" % (c, seed)
            assert t.start == tuple(g[k + "start"]) and t.goal == tuple(g[k + "goal"]), k
            assert np.array_equal(t.cell_walls, g[k + "walls"]), k
            assert np.array_equal(t.cell_texts, g[k + "texts"]), k
            assert np.array_equal(t.food_rewards, g[k + "food"]), k
            assert np.array_equal(t.food_interval, g[k + "interval"]), k
            assert np.array_equal(np.asarray([t.cell_size, t.wall_height, t.agent_height, t.initial_life,
                                              t.max_life, t.step_reward, t.goal_reward]), g[k + "scalars"]), k


def test_device_sampled_table_drives_envs_like_host_tasks(reference_textures):
    """A DeviceTaskTable handed to set_task gives the same episode as the same tasks uploaded from the
    host; seed_base + t addressing equals an explicit seed list; big tables are valid mazes."""
    import metagym_amd
    from metagym_amd.metamaze import MAZE_TASK_MANAGER
    kw = dict(n=9, allow_loops=False, step_reward=-0.01, goal_reward=1.0, food_density=0.06, food_interval=20)
    T, n_envs = 16, 64
    table = MAZE_TASK_MANAGER.sample_tasks_device(T, device="cuda:0", seed=100, **kw)
    table2 = MAZE_TASK_MANAGER.sample_tasks_device(T, device="cuda:0", seeds=list(range(100, 100 + T)), **kw)
    for k in table.tensors:
        assert torch.equal(table.tensors[k], table2.tensors[k]), k
    mk = lambda: metagym_amd.make("meta-maze-discrete-3D-v0", num_envs=n_envs, device="cuda:0", max_steps=40,
                                  resolution=(32, 32), task_type="SURVIVAL")
    a, b = mk(), mk()
    a.set_task(table)
    b.set_task(table.to_task_configs())
    oa, ob = a.reset(), b.reset()
    assert torch.equal(oa, ob)
    rs = np.random.RandomState(0)
    for _ in range(30):
        act = torch.as_tensor(rs.randint(0, 4, n_envs).astype(np.int32)).cuda()
        ra, rb = a.step(act), b.step(act)
        assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1]) and torch.equal(ra[2], rb[2])
    # 4096 tasks of the C1 shape: every maze is one connected corridor system containing start and goal
    big = MAZE_TASK_MANAGER.sample_tasks_device(4096, device="cuda:0", seed=0, n=15, allow_loops=True,
                                                crowd_ratio=0.35, step_reward=-0.01, goal_reward=1.0)
    w = big.tensors["walls"].cpu().numpy().reshape(4096, 15, 15)
    st, go = big.tensors["start"].cpu().numpy(), big.tensors["goal"].cpu().numpy()
    assert (w[:, 0, :] == 1).all() and (w[:, -1, :] == 1).all() and (w[:, :, 0] == 1).all() and (w[:, :, -1] == 1).all()
    for t in range(0, 4096, 97):
        free = w[t] == 0
        assert free[st[t, 0], st[t, 1]] and free[go[t, 0], go[t, 1]]
        seen = np.zeros_like(free)
        stack = [tuple(st[t])]
        seen[tuple(st[t])] = True
        while stack:
            i, j = stack.pop()
            for di, dj in ((1, 0), (-1, 0), (0, 1), (0, -1)):
                p = (i + di, j + dj)
                if free[p] and not seen[p]:
                    seen[p] = True
                    stack.append(p)
        assert seen.sum() == free.sum(), "maze %d is not connected" % t
        assert free[1:-1, 1:-1].size - free[1:-1, 1:-1].sum() <= 13 * 13 * 0.35
