"""oracle/maze_sampler.py — TEST INFRASTRUCTURE (CPU restatement; never imported by the product path).

Restates `MazeTaskManager.sample_task` (/root/reference metagym/metamaze/envs/maze_task.py:41-190)
INCLUDING the random streams it consumes, so that the device-side task sampler
(`mg_maze_sample_tasks`) has a bit-exact oracle:

    task(seed) := { random.seed(seed); numpy.random.seed(seed); sample_task(**kw) }

which is how SURVEY.md §8(d) defines the C1/C3 inputs and how oracle/gen_golden_maze.py draws the
tasks stored in the golden files. Two MT19937 generators are involved:

  * python's `random` (maze_task.py:71-72,79-80 randint; :103 shuffle; :137 random):
      seed(int)      -> init_by_array(32-bit little-endian words of |seed|)   (CPython _randommodule.c)
      _randbelow(n)  -> k = n.bit_length(); r = getrandbits(k) until r < n    (Lib/random.py)
      getrandbits(k) -> genrand_uint32() >> (32 - k)                          (k <= 32)
      random()       -> ((a >> 5) * 2^26 + (b >> 6)) / 2^53
      shuffle(x)     -> for i = len-1 .. 1: j = _randbelow(i + 1); swap
  * numpy's legacy global RandomState (maze_task.py:61 randint; :164,168 rand):
      seed(int)      -> init_genrand(seed)                                     (numpy _mt19937.pyx _legacy_seeding)
      randint(lo,hi) -> lo + masked rejection: (next_uint32 & mask) until <= hi-1-lo   (_bounded_integers, masked)
      rand()         -> ((a >> 5) * 2^26 + (b >> 6)) / 2^53

`numpy.sum` over the float64 food array (maze_task.py:167) is numpy's pairwise summation
(numpy/core/src/umath/loops_utils.h pairwise_sum): restated in `_pw` / `np_sum_f64` because the `while`
loop count depends on the exact rounded sum.

Pinned by tests/test_oracle_maze_sampler.py against tests/golden/maze_tasks.npz (tasks drawn by the
unmodified reference, oracle/gen_golden_maze_tasks.py) and against the task_* fields of every
maze golden file.
"""
import math
from collections import namedtuple

import numpy as np

TaskConfig = namedtuple("TaskConfig", ["start", "goal", "cell_walls", "cell_texts", "cell_size", "wall_height",
                                       "agent_height", "initial_life", "max_life", "step_reward", "goal_reward",
                                       "food_rewards", "food_interval"])

M32 = 0xFFFFFFFF


class MT19937(object):
    """Matsumoto & Nishimura's reference generator (mt19937ar.c)."""
    N, M = 624, 397

    def __init__(self):
        self.mt = [0] * self.N
        self.idx = self.N + 1

    def init_genrand(self, s):
        mt = self.mt
        mt[0] = s & M32
        for i in range(1, self.N):
            mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & M32
        self.idx = self.N

    def init_by_array(self, key):
        self.init_genrand(19650218)
        mt, N = self.mt, self.N
        i, j = 1, 0
        for _ in range(max(N, len(key))):
            mt[i] = ((mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525)) + key[j] + j) & M32
            i += 1
            j += 1
            if i >= N:
                mt[0] = mt[N - 1]
                i = 1
            if j >= len(key):
                j = 0
        for _ in range(N - 1):
            mt[i] = ((mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941)) - i) & M32
            i += 1
            if i >= N:
                mt[0] = mt[N - 1]
                i = 1
        mt[0] = 0x80000000
        self.idx = N

    def next_u32(self):
        mt, N, M = self.mt, self.N, self.M
        if self.idx >= N:
            for k in range(N):
                y = (mt[k] & 0x80000000) | (mt[(k + 1) % N] & 0x7FFFFFFF)
                mt[k] = mt[(k + M) % N] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
            self.idx = 0
        y = mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & M32

    def next_double(self):
        a, b = self.next_u32() >> 5, self.next_u32() >> 6
        return (a * 67108864.0 + b) / 9007199254740992.0


class PyRandom(object):
    """python's `random` module functions used by the sampler."""

    def __init__(self, seed):
        a = abs(int(seed))
        key = []
        while True:
            key.append(a & M32)
            a >>= 32
            if a == 0:
                break
        self.g = MT19937()
        self.g.init_by_array(key)

    def randbelow(self, n):
        k = n.bit_length()
        r = self.g.next_u32() >> (32 - k)
        while r >= n:
            r = self.g.next_u32() >> (32 - k)
        return r

    def randint(self, a, b):
        return a + self.randbelow(b - a + 1)

    def random(self):
        return self.g.next_double()

    def shuffle(self, x):
        for i in range(len(x) - 1, 0, -1):
            j = self.randbelow(i + 1)
            x[i], x[j] = x[j], x[i]


class NpLegacy(object):
    """numpy.random.seed / randint / rand of the legacy global RandomState."""

    def __init__(self, seed):
        self.g = MT19937()
        self.g.init_genrand(int(seed))

    def randint(self, low, high, count):
        rng = high - 1 - low
        if rng == 0:
            return [low] * count
        mask = rng
        for s in (1, 2, 4, 8, 16):
            mask |= mask >> s
        out = []
        for _ in range(count):
            v = self.g.next_u32() & mask
            while v > rng:
                v = self.g.next_u32() & mask
            out.append(low + v)
        return out

    def rand(self, count):
        return [self.g.next_double() for _ in range(count)]


def _pw(a, lo, n):
    if n < 8:
        res = 0.0
        for i in range(n):
            res = res + a[lo + i]
        return res
    if n <= 128:
        r = [a[lo + k] for k in range(8)]
        i = 8
        while i < n - (n % 8):
            for k in range(8):
                r[k] = r[k] + a[lo + i + k]
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < n:
            res = res + a[lo + i]
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return _pw(a, lo, n2) + _pw(a, lo + n2, n - n2)


def np_sum_f64(a):
    """numpy.sum of a C-contiguous float64 array: the reduce loop sees one run of a.size elements and
    computes 0.0 + pairwise_sum(run) (checked against numpy 2.2 on 3000 random arrays)."""
    a = [float(x) for x in np.asarray(a, np.float64).reshape(-1)]
    return 0.0 + _pw(a, 0, len(a))


def sample_task(seed, n_texts, n=15, allow_loops=True, cell_size=2.0, wall_height=3.2, agent_height=1.6,
                step_reward=-0.01, goal_reward=None, food_reward=0.50, initial_life=1.0, max_life=2.0,
                food_density=0.010, food_interval=100, crowd_ratio=0.0):
    """maze_task.py:41-190 after `random.seed(seed); numpy.random.seed(seed)`. n_texts = number of
    ground/wall textures of the manager (maze_task.py:38-39; 7 for the shipped texture set)."""
    assert n > 6, "Minimum required cells are 7"
    assert n % 2 != 0, "Cell Numbers can only be odd"
    py, npr = PyRandom(seed), NpLegacy(seed)
    walls = [[1] * n for _ in range(n)]
    texts = npr.randint(1, n_texts, n * n)                                        # :61
    texts = [texts[i * n:(i + 1) * n] for i in range(n)]
    for i in range(1, n, 2):                                                      # :64-66
        for j in range(1, n, 2):
            walls[i][j] = 0
    m = (n - 1) // 2
    s_x = py.randint(0, m - 1) * 2 + 1                                            # :68-69
    s_y = py.randint(0, m - 1) * 2 + 1
    goal = (n - 2, n - 2)
    min_dist = 0.45 * n
    for _ in range(1, n, 2):                                                      # :75-83 (`break` leaves the inner loop only)
        for _ in range(1, n, 2):
            e_x = py.randint(0, m - 1) * 2 + 1
            e_y = py.randint(0, m - 1) * 2 + 1
            if math.sqrt((e_x - s_x) ** 2 + (e_y - s_y) ** 2) > min_dist:
                goal = (e_x, e_y)
                break
    wall_list, path = [], {}                                                      # :86-98
    n_paths = 0
    for i in range(1, n - 1):
        for j in range(1, n - 1):
            if walls[i][j] > 0:
                wall_list.append((i, j))
            else:
                path[i, j] = n_paths
                n_paths += 1
    n_wall_cells = len(wall_list)        # == sum(cell_walls[1:-1, 1:-1])
    max_cell_walls = (n - 2) * (n - 2)
    while n_paths > 1 or (allow_loops and n_wall_cells > max_cell_walls * crowd_ratio):   # :103
        order = list(wall_list)
        py.shuffle(order)
        new_id = -1
        abandon = []
        i = j = -1
        for i, j in order:
            new_id, abandon, count, max_dup = -1, [], {}, 1
            for d_i, d_j in ((i - 1, j), (i + 1, j), (i, j - 1), (i, j + 1)):
                if 0 < d_i < n and 0 < d_j < n and walls[d_i][d_j] < 1:
                    pid = path[d_i, d_j]
                    count[pid] = count.get(pid, 0) + 1
                    max_dup = max(max_dup, count[pid])
                    if pid < new_id or new_id < 0:
                        if new_id >= 0 and new_id not in abandon:
                            abandon.append(new_id)
                        new_id = pid
                    elif pid != new_id and pid not in abandon:
                        abandon.append(pid)
            if abandon and max_dup < 2:
                break
            if abandon and max_dup > 1 and allow_loops:
                break
            if allow_loops and n_paths < 2 and py.random() < 0.2:
                break
        if new_id < 0:
            continue
        path[i, j] = new_id                                                       # :144-154
        walls[i][j] = 0
        wall_list.remove((i, j))
        n_wall_cells -= 1
        for pid in abandon:
            for k in path:
                if path[k] == pid:
                    path[k] = new_id
            n_paths -= 1
    for i in range(1, n - 1):                                                     # :157-160
        for j in range(1, n - 1):
            if walls[i][j] < 1:
                texts[i][j] = 0
    assert step_reward < 0, "step_reward must be < 0"
    def_goal_reward = -math.sqrt(n) * n * step_reward if goal_reward is None else goal_reward
    assert def_goal_reward > 0, "goal reward must be > 0"
    walls_a = np.asarray(walls, np.int32)
    food = np.clip(np.asarray(npr.rand(n * n)).reshape(n, n) * food_reward, 0.10, food_reward)   # :164
    food *= 1.0 - walls_a
    exp_food = (n - 1) * (n - 1) * food_density
    while np_sum_f64(food) > exp_food:                                            # :167-168
        food *= (np.asarray(npr.rand(n * n)).reshape(n, n) < 0.90).astype("float32")
    interval = food_interval * (food > 1.0e-3).astype("int32")
    return TaskConfig(start=(s_x, s_y), goal=goal, cell_walls=walls_a, cell_texts=np.asarray(texts, np.int64),
                      cell_size=cell_size, step_reward=step_reward, goal_reward=def_goal_reward,
                      wall_height=wall_height, agent_height=agent_height, initial_life=initial_life,
                      max_life=max_life, food_rewards=food, food_interval=interval)
