"""Host-side task generation for MetaMaze (the reference's L0 layer, metagym/metamaze/envs/maze_task.py).

`TaskConfig` has exactly the reference's fields (maze_task.py:15-17) so task objects are
interchangeable: a TaskConfig sampled by the reference can be handed to these envs and vice versa.

`MazeTaskSampler` is the reference's generator (maze_task.py:41-190) restated: for the same python-`random`
and `numpy.random` streams it returns the reference's task, field for field (see `sample_task`); the same
tasks can be drawn on the GPU with `sample_tasks_device` (`mg_maze_sample_tasks`).

Textures: the reference ships seven 64x64 PNGs. `MazeTaskManager(texture_dir=...)` loads any
directory with the same naming rule (file names containing 'ground' / 'wall' / 'ceil', sorted);
without one, deterministic procedural textures of the same shape are generated.
"""
import os
from collections import namedtuple

import numpy as np

TaskConfig = namedtuple("TaskConfig", ["start", "goal", "cell_walls", "cell_texts", "cell_size", "wall_height",
                                       "agent_height", "initial_life", "max_life", "step_reward", "goal_reward",
                                       "food_rewards", "food_interval"])


class DeviceTaskTable(object):
    """T tasks of one size resident on the GPU — the arrays `mg_maze_tasks` points at. Produced by
    `MazeTaskManager.sample_tasks_device` (no host round trip) and accepted by `env.set_task`."""

    KEYS = ("start", "goal", "walls", "texts", "food_rewards", "food_interval", "scalars")

    def __init__(self, n, tensors, cell_size):
        self.n = int(n)
        self.cell_size = float(cell_size)      # uniform over the table (a sampler parameter)
        self.tensors = tensors
        self.n_tasks = int(tensors["start"].shape[0])

    def __len__(self):
        return self.n_tasks

    def to_task_configs(self):
        """Copy to host as reference-style TaskConfig tuples (inspection / tests)."""
        h = {k: v.cpu().numpy() for k, v in self.tensors.items()}
        n, out = self.n, []
        for t in range(self.n_tasks):
            sc = h["scalars"][t]
            out.append(TaskConfig(start=tuple(int(x) for x in h["start"][t]), goal=tuple(int(x) for x in h["goal"][t]),
                                  cell_walls=h["walls"][t].reshape(n, n).astype(np.int32),
                                  cell_texts=h["texts"][t].reshape(n, n).astype(np.int64),
                                  cell_size=float(sc[0]), wall_height=float(sc[1]), agent_height=float(sc[2]),
                                  initial_life=float(sc[3]), max_life=float(sc[4]), step_reward=float(sc[5]),
                                  goal_reward=float(sc[6]), food_rewards=h["food_rewards"][t].reshape(n, n),
                                  food_interval=h["food_interval"][t].reshape(n, n)))
        return out


def _procedural_textures(n_walls=6, size=64):
    rs = np.random.RandomState(20260925)
    yy, xx = np.mgrid[0:size, 0:size]

    def noise(scale):
        return rs.randint(-scale, scale + 1, size=(size, size, 1))

    texs = []
    ground = 120 + noise(25) + ((xx // 16 + yy // 16) % 2 * 14)[..., None]
    texs.append(np.repeat(ground, 3, axis=2))
    palettes = [(150, 60, 50), (60, 80, 160), (130, 130, 120), (70, 120, 60), (170, 90, 60), (120, 60, 110)]
    for k in range(n_walls):
        base = np.array(palettes[k % len(palettes)])[None, None, :]
        brick_h, brick_w = 8 + 2 * (k % 3), 16 + 4 * (k % 2)
        row = yy // brick_h
        mortar = ((yy % brick_h) == 0) | (((xx + (row % 2) * (brick_w // 2)) % brick_w) == 0)
        t = base + noise(18)
        t = np.where(mortar[..., None], 200 + noise(10), t)
        texs.append(t)
    grounds = np.clip(np.stack(texs), 0, 252).astype(np.uint8)
    ceil = np.clip(np.array([110, 80, 50])[None, None, :] + noise(12) + ((xx // 8) % 2 * 18)[..., None], 0, 255)
    return grounds, ceil.astype(np.uint8)


def _load_texture_dir(texture_dir):
    """Same selection rule as MazeTaskManager.__init__ (maze_task.py:19-36); images are returned in
    pygame's surfarray convention (axis 0 = x)."""
    from PIL import Image

    def load(path):
        im = Image.open(path).convert("RGB")
        return np.transpose(np.asarray(im, dtype=np.uint8), (1, 0, 2)).copy()

    grounds, ceil = [None], None
    for name in sorted(os.listdir(texture_dir)):
        path = os.path.join(texture_dir, name)
        if name.find("wall") >= 0:
            grounds.append(load(path))
        if name.find("ground") >= 0:
            grounds[0] = load(path)
        if name.find("ceil") >= 0:
            ceil = load(path)
    if grounds[0] is None or ceil is None or len(grounds) < 2:
        raise ValueError("texture dir %r needs ground*, wall* and ceil* images" % texture_dir)
    return np.stack(grounds).astype(np.uint8), ceil


class MazeTaskManager(object):
    TaskConfig = TaskConfig

    def __init__(self, texture_dir=None, verbose=False):
        if texture_dir is None:
            texture_dir = os.environ.get("METAGYM_MAZE_TEXTURES")
        if texture_dir:
            g, c = _load_texture_dir(texture_dir)
        else:
            g, c = _procedural_textures()
        self.set_textures(g, c)
        self.verbose = verbose

    def set_textures(self, grounds_u8, ceil_u8):
        """grounds [n_texts, S, S, 3] uint8 (index 0 = floor), ceil [S, S, 3] uint8."""
        g = np.ascontiguousarray(grounds_u8, dtype=np.uint8)
        c = np.ascontiguousarray(ceil_u8, dtype=np.uint8)
        assert g.ndim == 4 and g.shape[1] == g.shape[2] and g.shape[3] == 3
        assert c.shape == g.shape[1:]
        self.grounds = g.astype(np.float32)     # the reference keeps float32 (maze_task.py:36)
        self.ceil = c
        self.version = getattr(self, "version", 0) + 1

    @property
    def n_texts(self):
        return self.grounds.shape[0]

    def packed_textures(self):
        """uint32 texels r | g<<8 | b<<16 for the device: ([n_texts,S,S], [S,S])."""
        g = self.grounds.astype(np.uint32)
        c = self.ceil.astype(np.uint32)
        return (g[..., 0] | (g[..., 1] << 8) | (g[..., 2] << 16)).astype(np.uint32), \
               (c[..., 0] | (c[..., 1] << 8) | (c[..., 2] << 16)).astype(np.uint32)

    def seed(self, seed=None):
        """Seed BOTH global streams the generator draws from, like a reference user would
        (`random.seed(seed); numpy.random.seed(seed)`)."""
        import random as _pyrandom
        _pyrandom.seed(seed)
        np.random.seed(seed)

    def sample_task(self, n=15, allow_loops=True, cell_size=2.0, wall_height=3.2, agent_height=1.6,
                    step_reward=-0.01, goal_reward=None, food_reward=0.50, initial_life=1.0, max_life=2.0,
                    food_density=0.010, food_interval=100, crowd_ratio=0.0, seed=None):
        """The reference's task generator (`MazeTaskManager.sample_task`, maze_task.py:41-190), same signature
        plus `seed`, and the same task for the same random streams:

          * `seed=None`: draws from python's global `random` and `numpy.random`, exactly like the reference —
            `random.seed(s); numpy.random.seed(s); MazeTaskSampler(...)` returns the task the reference returns;
          * `seed=s`: the same task from private generators (`random.Random(s)`, `numpy.random.RandomState(s)`),
            leaving the global streams alone. `sample_tasks_device(..., seed=s)` row 0 is this task, drawn on
            the GPU.

        The algorithm is the reference's: texture ids for every cell (:61), corridors on the odd lattice (:64-66),
        start and goal from python's generator (:68-83), then walls are dug one at a time — the wall list is
        re-shuffled, walked until a wall qualifies (it joins two different components; or, with loops allowed, it
        closes a cycle next to a component boundary, or a coin flip once everything is connected) and removed,
        merging the components it touches — until one component is left and, with `allow_loops`, the interior wall
        share is down to `crowd_ratio` (:103-154); passages get texture 0 (:157-160); food is drawn and thinned by
        10 % sweeps until its sum fits `food_density` (:163-169). The draw order of both streams is part of the
        contract; tests/test_maze_host_sampler.py checks 96 tasks recorded from the unmodified reference."""
        import math
        import random as _pyrandom
        assert n > 6, "Minimum required cells are 7"
        assert n % 2 != 0, "Cell Numbers can only be odd"
        py = _pyrandom if seed is None else _pyrandom.Random(seed)
        npr = np.random if seed is None else np.random.RandomState(seed)

        walls = np.ones((n, n), dtype="int32")
        texts = npr.randint(1, self.n_texts, size=(n, n))
        walls[1:n:2, 1:n:2] = 0
        half = (n - 1) // 2
        start = (py.randint(0, half - 1) * 2 + 1, py.randint(0, half - 1) * 2 + 1)
        goal, far_enough = (n - 2, n - 2), 0.45 * n
        for _row in range(half):               # a hit only ends the inner sweep (the reference's `break`), so up
            for _col in range(half):           # to `half` goals are drawn and the last one found is kept
                cand = (py.randint(0, half - 1) * 2 + 1, py.randint(0, half - 1) * 2 + 1)
                if math.sqrt((cand[0] - start[0]) ** 2 + (cand[1] - start[1]) ** 2) > far_enough:
                    goal = cand
                    break

        # component label of every open interior cell (row-major numbering), and the diggable walls in the same order
        label = {}
        candidates = []
        for i in range(1, n - 1):
            for j in range(1, n - 1):
                if walls[i, j] > 0:
                    candidates.append((i, j))
                else:
                    label[i, j] = len(label)
        n_components = len(label)
        walls_left = len(candidates)
        wall_budget = (n - 2) * (n - 2) * crowd_ratio
        while n_components > 1 or (allow_loops and walls_left > wall_budget):
            order = list(candidates)
            py.shuffle(order)
            pick, keep, absorbed = None, -1, []
            for pick in order:
                keep, absorbed, seen, twice = -1, [], {}, False
                i, j = pick
                for a, b in ((i - 1, j), (i + 1, j), (i, j - 1), (i, j + 1)):
                    if 0 < a < n and 0 < b < n and walls[a, b] < 1:
                        c = label[a, b]
                        seen[c] = seen.get(c, 0) + 1
                        twice = twice or seen[c] > 1
                        if keep < 0 or c < keep:         # the smallest label survives a merge
                            if keep >= 0 and keep not in absorbed:
                                absorbed.append(keep)
                            keep = c
                        elif c != keep and c not in absorbed:
                            absorbed.append(c)
                if absorbed and (not twice or allow_loops):
                    break
                if allow_loops and n_components < 2 and py.random() < 0.2:
                    break
            if keep < 0:
                continue
            walls[pick] = 0
            label[pick] = keep
            candidates.remove(pick)
            walls_left -= 1
            for c in absorbed:
                for cell, lab in label.items():
                    if lab == c:
                        label[cell] = keep
                n_components -= 1

        inner = texts[1:n - 1, 1:n - 1]
        inner[walls[1:n - 1, 1:n - 1] < 1] = 0

        assert step_reward < 0, "step_reward must be < 0"
        def_goal_reward = -np.sqrt(n) * n * step_reward if goal_reward is None else goal_reward
        assert def_goal_reward > 0, "goal reward must be > 0"

        food = np.clip(npr.rand(n, n) * food_reward, 0.10, food_reward)
        food *= 1.0 - walls
        expected = (n - 1) * (n - 1) * food_density
        while np.sum(food) > expected:
            food *= (npr.rand(n, n) < 0.90).astype("float32")
        interval = food_interval * (food > 1.0e-3).astype("int32")
        return TaskConfig(start=start, goal=goal, cell_walls=walls, cell_texts=texts, cell_size=cell_size,
                          step_reward=step_reward, goal_reward=def_goal_reward, wall_height=wall_height,
                          agent_height=agent_height, initial_life=initial_life, max_life=max_life,
                          food_rewards=food, food_interval=interval)

    def sample_tasks_device(self, num_tasks, device="cuda", seed=0, seeds=None, n=15, allow_loops=True,
                            cell_size=2.0, wall_height=3.2, agent_height=1.6, step_reward=-0.01, goal_reward=None,
                            food_reward=0.50, initial_life=1.0, max_life=2.0, food_density=0.010, food_interval=100,
                            crowd_ratio=0.0):
        """`num_tasks` tasks drawn ON THE GPU by `mg_maze_sample_tasks` (one wave per task). Task t is
        bit-identical to the reference's
            random.seed(s_t); numpy.random.seed(s_t); MazeTaskSampler(n=..., ...)
        with s_t = seeds[t] (uint32 tensor / array) or seed + t. Returns a DeviceTaskTable for
        `env.set_task`; nothing is copied to the host."""
        import torch
        from .. import _lib
        lib = _lib.load()
        dev = torch.device(device)
        T, nn = int(num_tasks), int(n) * int(n)
        tens = dict(start=torch.empty(T, 2, dtype=torch.int32, device=dev),
                    goal=torch.empty(T, 2, dtype=torch.int32, device=dev),
                    walls=torch.empty(T, nn, dtype=torch.int8, device=dev),
                    texts=torch.empty(T, nn, dtype=torch.uint8, device=dev),
                    food_rewards=torch.empty(T, nn, dtype=torch.float64, device=dev),
                    food_interval=torch.empty(T, nn, dtype=torch.int32, device=dev),
                    scalars=torch.empty(T, 8, dtype=torch.float64, device=dev))
        p = _lib.MazeSampleParams()
        p.n, p.allow_loops, p.n_texts, p.food_interval = int(n), int(bool(allow_loops)), self.n_texts, int(food_interval)
        p.has_goal_reward = int(goal_reward is not None)
        p.cell_size, p.wall_height, p.agent_height = float(cell_size), float(wall_height), float(agent_height)
        p.step_reward, p.goal_reward = float(step_reward), float(goal_reward if goal_reward is not None else 0.0)
        p.food_reward, p.initial_life, p.max_life = float(food_reward), float(initial_life), float(max_life)
        p.food_density, p.crowd_ratio = float(food_density), float(crowd_ratio)
        seeds_t = None
        if seeds is not None:
            # uint32 values carried in an int64 -> int32-bit-pattern tensor (torch has no uint32 arithmetic)
            arr = np.asarray(seeds.cpu().numpy() if hasattr(seeds, "cpu") else seeds, dtype=np.int64)
            assert arr.shape == (T,) and arr.min() >= 0 and arr.max() < 2 ** 32, "seeds must be T values in [0, 2^32)"
            seeds_t = torch.from_numpy(arr.astype(np.uint32).view(np.int32)).to(dev)
        assert 0 <= int(seed) and int(seed) + T <= 2 ** 32, "seeds are 32-bit (numpy.random.seed's integer range)"
        rc = lib.mg_maze_sample_tasks(p, T, int(seed), _lib.ptr(seeds_t), *[_lib.ptr(tens[k]) for k in DeviceTaskTable.KEYS],
                                      _lib.current_stream(dev))
        _lib.check(rc, "mg_maze_sample_tasks")
        return DeviceTaskTable(n, tens, cell_size)


MAZE_TASK_MANAGER = MazeTaskManager()
MazeTaskSampler = MAZE_TASK_MANAGER.sample_task
