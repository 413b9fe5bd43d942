"""Batched MetaMaze envs — host-side mirror of metagym/metamaze/envs/maze_env.py:16,85,155.

Same class names, constructor arguments, `set_task()` / `reset()` / `step()` surface and the same
observation dtypes (2-D: float32 window; 3-D: int32 [H,V,3] image whose values exceed 255), but
every call advances `num_envs` mazes with one launch of the gfx950 kernels behind
`mg_maze2d_step` / `mg_maze3d_step` (include/metagym_hip.h, metagym_amd/csrc/maze.hip) and returns
torch-ROCm tensors of shape [num_envs, ...].

Batched extensions
  * `set_task(task)` gives every env the same TaskConfig; `set_task([t0, t1, ...], task_ids=ids)`
    uploads a table of T tasks (all the same size n) and env e plays task `ids[e]`
    (default e mod T).
  * the reference raises when step() is called on a finished episode (maze_env.py:60-61); per-lane
    exceptions do not exist on a GPU, so either reset the finished envs with `reset(mask=done)` or
    construct the env with `auto_reset=True` (finished envs restart inside the same launch and the
    returned observation is the first one of the next episode).
  * rendering (`enable_render`, pygame god-view) is out of scope; the argument is accepted and ignored.
"""

import numpy as np
import torch

from .. import _lib
from ..spaces import Box, Discrete
from .maze_task import MAZE_TASK_MANAGER, DeviceTaskTable, TaskConfig  # noqa: F401  (TaskConfig re-exported like the reference's maze_env)

PI = 3.1415926                      # dynamics.py:6
DISCRETE_ACTIONS = [(-1, 0), (1, 0), (0, -1), (0, 1)]   # maze_env.py:14
TASK_TYPES = {"ESCAPE": 0, "SURVIVAL": 1}


class _MazeBatch(object):
    """State, task table and launch plumbing shared by the three envs."""

    def __init__(self, num_envs, device, max_steps, task_type, auto_reset):
        assert task_type in TASK_TYPES, "task_type must be ESCAPE or SURVIVAL"
        self._lib = _lib.load()
        self.num_envs = int(num_envs)
        self.device = _lib.canonical_device(device)
        if self.device.type != "cuda":
            raise _lib.MetaGymHipError("metagym_amd runs on an AMD GPU only (got device %r); there is no CPU "
                                       "path" % (device,))
        self.max_steps = int(max_steps)
        self.task_type = task_type
        self._tt = TASK_TYPES[task_type]
        self.auto_reset = bool(auto_reset)
        self.need_set_task = True
        self.need_reset = True
        self._tasks_c = None
        N, dev = self.num_envs, self.device
        self._reward = torch.zeros(N, dtype=torch.float32, device=dev)
        self._reward64 = torch.zeros(N, dtype=torch.float64, device=dev)
        self._done = torch.zeros(N, dtype=torch.bool, device=dev)

    # ------------------------------------------------------------------ tasks
    def __del__(self):
        try:
            if self._tasks_c is not None:
                self._lib.mg_maze_forget_tasks(self._tasks_c)
        except Exception:
            pass

    def set_task(self, task_config, task_ids=None):
        """task_config: one TaskConfig, a list of T TaskConfigs (uploaded once), or a DeviceTaskTable from
        `MazeTaskManager.sample_tasks_device` (already on the GPU). Env e plays task task_ids[e]
        (default e mod T)."""
        N, dev = self.num_envs, self.device
        if isinstance(task_config, DeviceTaskTable):
            n, T = task_config.n, task_config.n_tasks
            assert all(v.device == dev or (v.device.type == dev.type and (v.device.index or 0) == (dev.index or 0))
                       for v in task_config.tensors.values()), "task table lives on another device"
            self._task_t = task_config.tensors
            self.tasks = task_config
            self._min_cell_size = float(task_config.cell_size)
            self._uniform_cell_size = float(task_config.cell_size)       # one sampler configuration: one cell size
            host = task_config.tensors
        else:
            tasks = [task_config] if isinstance(task_config, tuple) and hasattr(task_config, "cell_walls") else list(task_config)
            assert len(tasks) >= 1
            n = int(np.shape(tasks[0].cell_walls)[0])
            T = len(tasks)
            for t in tasks:   # maze_base.py:35-38
                assert np.shape(t.cell_walls) == (n, n), "all tasks of one batch must have the same size"
                assert 0 < t.agent_height < t.wall_height, "the agent height must be > 0 and < wall height"
                assert np.shape(t.cell_walls) == np.shape(t.cell_texts), "the dimension of walls must be equal to textures"
            host = dict(
                start=np.asarray([t.start for t in tasks], np.int32).reshape(T, 2),
                goal=np.asarray([t.goal for t in tasks], np.int32).reshape(T, 2),
                walls=np.clip(np.asarray([t.cell_walls for t in tasks]), -1, 1).astype(np.int8).reshape(T, n * n),
                texts=np.asarray([t.cell_texts for t in tasks]).astype(np.uint8).reshape(T, n * n),
                food_rewards=np.asarray([t.food_rewards for t in tasks], np.float64).reshape(T, n * n),
                food_interval=np.asarray([t.food_interval for t in tasks], np.int32).reshape(T, n * n),
                scalars=np.asarray([[t.cell_size, t.wall_height, t.agent_height, t.initial_life, t.max_life,
                                     t.step_reward, t.goal_reward, 0.0] for t in tasks], np.float64))
            assert int(host["texts"].max()) < MAZE_TASK_MANAGER.n_texts, "cell_texts refers to a missing texture"
            self._task_t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in host.items()}
            self.tasks = tasks
            self._min_cell_size = min(float(t.cell_size) for t in tasks)
            sizes = set(float(t.cell_size) for t in tasks)
            self._uniform_cell_size = sizes.pop() if len(sizes) == 1 else 0.0   # mg_maze_view.uniform_cell_size: 0 = tasks differ
        nn = n * n
        self.n = n
        if self._tasks_c is not None:
            # the table this env held goes away (its tensors are released below): the library's address-keyed memory of a
            # checked (table, uniform_cell_size) pair must not outlive it — torch's caching allocator reuses the address
            self._lib.mg_maze_forget_tasks(self._tasks_c)
        c = _lib.MazeTasks()
        c.n, c.n_tasks = n, T
        for k in host:
            setattr(c, k, self._task_t[k].data_ptr())
        # compact list of the cells that can ever hold food (exact shortcut for the SURVIVAL sweeps, see
        # include/metagym_hip.h); built with device ops so a DeviceTaskTable never visits the host
        tt_food, tt_int = self._task_t["food_rewards"], self._task_t["food_interval"]
        can = (tt_int > 0) | (tt_food > 1.0e-2)                                  # [T, nn]
        n_food = can.sum(dim=1).to(torch.int32)
        max_food = max(1, int(n_food.max().item()))
        order = torch.argsort((~can).to(torch.int8), dim=1, stable=True)[:, :max_food]   # food cells first, ascending
        self._food_cells_t = order.to(torch.int16).contiguous()
        self._n_food_t = n_food.contiguous()
        c.food_cells, c.n_food, c.max_food = self._food_cells_t.data_ptr(), self._n_food_t.data_ptr(), max_food
        self._max_food = max_food
        if self._BY_SLOT:
            # the list once more for the lane-per-env kernel (mg_maze_tasks.cell_slot / slot_food / slot_interval): the inverse
            # map cell -> slot, and the listed cells' task values slot-major, [max_food][T]
            listed = torch.arange(max_food, device=dev).unsqueeze(0) < n_food.unsqueeze(1)              # [T, max_food]
            # cells outside the list: -1 when the task's value there is exactly 0.0 (the kernel then reads nothing for them), -2 when
            # it is a nonzero value <= 1e-2 (never eaten, never renewed, but part of the 2-D observation: read from the task)
            slot = torch.where(tt_food != 0.0, torch.full((T, nn), -2, dtype=torch.int16, device=dev),
                               torch.full((T, nn), -1, dtype=torch.int16, device=dev))
            ks = torch.arange(max_food, device=dev, dtype=torch.int16).unsqueeze(0).expand(T, max_food)
            # only the LISTED entries of a row are written: a task with n_food < max_food has non-food cells in the tail of its
            # `order` row, and those keep their -1 / -2 (a -1 there would hide a nonzero crumb from the observation)
            slot.scatter_(1, order, torch.where(listed, ks, slot.gather(1, order)))        # (a row of `order` holds distinct cells)
            self._cell_slot_t = slot.contiguous()
            self._slot_food_t = torch.gather(tt_food, 1, order).t().contiguous()
            self._slot_interval_t = torch.gather(tt_int, 1, order).t().contiguous()
            c.cell_slot, c.slot_food, c.slot_interval = (self._cell_slot_t.data_ptr(), self._slot_food_t.data_ptr(),
                                                         self._slot_interval_t.data_ptr())
        self._tasks_c = c
        if self._uniform_cell_size > 0.0:
            # the promise mg_maze_view.uniform_cell_size makes about THIS table, checked by the library against the uploaded
            # cell_size column (synchronous, once per set_task; a hipGraph capture of step() later finds the pair checked)
            _lib.check(self._lib.mg_maze_check_uniform_cell_size(c, self._uniform_cell_size, _lib.current_stream(dev)),
                       "mg_maze_check_uniform_cell_size")
        if task_ids is None:
            task_ids = torch.arange(N, dtype=torch.int32) % T
        self.task_id = torch.as_tensor(task_ids, dtype=torch.int32).to(dev).contiguous()
        assert self.task_id.shape == (N,)
        # ---- per-env state --------------------------------------------------------------------
        self.grid = torch.zeros(2, N, dtype=torch.int32, device=dev)
        self.steps = torch.zeros(N, dtype=torch.int32, device=dev)
        self.ori_idx = torch.zeros(N, dtype=torch.int32, device=dev)
        self.ori = torch.zeros(N, dtype=torch.float64, device=dev)
        self.loc = torch.zeros(2, N, dtype=torch.float32, device=dev)
        self.life = torch.zeros(N, dtype=torch.float64, device=dev)
        st = _lib.MazeState()
        st.task_id, st.grid, st.steps = self.task_id.data_ptr(), self.grid.data_ptr(), self.steps.data_ptr()
        st.ori_idx, st.ori, st.loc, st.life = (self.ori_idx.data_ptr(), self.ori.data_ptr(), self.loc.data_ptr(),
                                               self.life.data_ptr())
        if self._tt == TASK_TYPES["SURVIVAL"]:
            # [N, nn] cells for the workgroup-per-env 3-D kernel; [max_food, N] food SLOTS for the lane-per-env 2-D kernel
            # (mg_maze_state.food_by_slot: lane e's k-th access is coalesced whatever task it runs)
            shape = (self._max_food, N) if self._BY_SLOT else (N, nn)
            self.cur_food = torch.zeros(*shape, dtype=torch.float64, device=dev)
            self.wait_refresh = torch.zeros(*shape, dtype=torch.uint8, device=dev)
            self.revival = torch.zeros(*shape, dtype=torch.int32, device=dev)
            st.cur_food, st.wait_refresh, st.revival = (self.cur_food.data_ptr(), self.wait_refresh.data_ptr(),
                                                        self.revival.data_ptr())
            st.food_env_stride, st.food_cell_stride = (1, N) if self._BY_SLOT else (nn, 1)
            st.food_by_slot = int(self._BY_SLOT)
        self._state_c = st
        self._on_set_task()
        self.need_set_task = False
        self.need_reset = True

    def _on_set_task(self):
        pass

    _BY_SLOT = False
    _STATE_KEYS = ("grid", "steps", "ori_idx", "ori", "loc", "life", "cur_food", "wait_refresh", "revival")
    _FOOD_KEYS = ("cur_food", "wait_refresh", "revival")

    def _food_by_cell(self, key):
        """(slot layout) the [n*n, N] by-cell view of a SURVIVAL array — the checkpoint format, independent of how the kernel
        stores it: a listed cell's slot value, else what the cell holds for ever (its task's food value, wait 0, counter =
        its interval)."""
        tid = self.task_id.long()
        slot = self._cell_slot_t[tid].t().long()                                   # [nn, N]
        have = slot >= 0
        val = torch.gather(getattr(self, key), 0, slot.clamp(min=0))
        if key == "cur_food":
            rest = self._task_t["food_rewards"][tid].t()
        elif key == "revival":
            rest = self._task_t["food_interval"][tid].t()
        else:
            rest = torch.zeros_like(val)
        return torch.where(have, val, rest.to(val.dtype)).contiguous()

    def state_dict(self):
        """Per-env arrays + task_id: the SURVIVAL food arrays only make sense next to the task they were drawn for. The food
        arrays are always written by CELL ([n*n, N] for the 2-D env, [N, n*n] for the 3-D ones)."""
        sd = {}
        for k in self._STATE_KEYS:
            if hasattr(self, k):
                sd[k] = self._food_by_cell(k) if (self._BY_SLOT and k in self._FOOD_KEYS) else getattr(self, k).clone()
        if hasattr(self, "task_id"):
            sd["task_id"] = self.task_id.clone()
        return sd

    def load_state_dict(self, sd):
        if "task_id" in sd and hasattr(self, "task_id"):          # first: the by-cell food arrays are read through the task's slots
            self.task_id.copy_(torch.as_tensor(sd["task_id"]).to(self.task_id.dtype))
        for k in self._STATE_KEYS:
            if hasattr(self, k) and k in sd:
                dst, src = getattr(self, k), torch.as_tensor(sd[k]).to(self.device)
                if self._BY_SLOT and k in self._FOOD_KEYS:
                    if tuple(src.shape) != (self.n * self.n, self.num_envs):
                        raise ValueError("state_dict[%r] has shape %s, expected the by-cell view %s" % (k, tuple(src.shape), (self.n * self.n, self.num_envs)))
                    cells = self._food_cells_t[self.task_id.long()].t().long()   # [max_food, N]: the cell behind every slot
                    dst.copy_(torch.gather(src, 0, cells).to(dst.dtype))           # (slots past a task's n_food are never read)
                    continue
                if tuple(src.shape) != tuple(dst.shape):
                    raise ValueError("state_dict[%r] has shape %s, this env holds %s (same num_envs and maze size n needed)"
                                     % (k, tuple(src.shape), tuple(dst.shape)))
                dst.copy_(src.to(dst.dtype))
        if self._BY_SLOT and hasattr(self, "revival") and all(k in sd for k in ("revival", "wait_refresh")):
            # the slot kernel never reads the counter of a slot that is not waiting (mg_maze_state.food_by_slot): a state with a
            # negative counter there cannot come out of reset() / step() — refuse it instead of diverging from maze_base.py:83-88
            if bool(((self.revival < 0) & (self.wait_refresh == 0)).any()):
                raise ValueError("state_dict: a food counter below zero on a cell that is not waiting (unreachable by reset / step)")

    # ------------------------------------------------------------------ episode control
    def reset(self, mask=None):
        if self.need_set_task:
            raise Exception("Must call \"set_task\" before reset")          # maze_env.py:49-50
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
            assert m.shape == (self.num_envs,)
        rc = self._lib.mg_maze_reset(self._tasks_c, self._tt, self.num_envs, self._state_c, _lib.ptr(m),
                                     _lib.current_stream(self.device))
        _lib.check(rc, "mg_maze_reset")
        self.need_reset = False
        return self._observe()

    def _check_step(self):
        if self.need_reset:
            raise Exception("Must \"reset\" before doing any actions")        # maze_env.py:60-61

    @property
    def reward64(self):
        return self._reward64

    def render(self, mode="human"):
        raise NotImplementedError("rendering is out of scope for the batched engine")

    def close(self):
        pass


class MetaMaze2D(_MazeBatch):
    """maze_env.py:155-212. obs float32 [N, 2v+1, 2v+1]; action int in {0..3} per env."""
    _BY_SLOT = True       # SURVIVAL arrays per food slot, [max_food, N] (mg_maze_state.food_by_slot)

    def __init__(self, num_envs=1, device="cuda", enable_render=False, render_scale=480, max_steps=5000,
                 task_type="SURVIVAL", view_grid=2, auto_reset=False):
        super().__init__(num_envs, device, max_steps, task_type, auto_reset)
        self.view_grid = int(view_grid)
        self.action_space = Discrete(4)
        w = 2 * self.view_grid + 1
        self.observation_space = Box(low=-1.0, high=2.0, shape=(w, w), dtype=np.float32)
        self._obs = torch.zeros(self.num_envs, w, w, dtype=torch.float32, device=self.device)

    def _launch(self, action):
        rc = self._lib.mg_maze2d_step(self._tasks_c, self._tt, self.max_steps, self.view_grid, int(self.auto_reset),
                                      self.num_envs, self._state_c, _lib.ptr(action), _lib.ptr(self._obs),
                                      _lib.ptr(self._reward), _lib.ptr(self._reward64), _lib.ptr(self._done),
                                      _lib.current_stream(self.device))
        _lib.check(rc, "mg_maze2d_step")

    def _observe(self):
        self._launch(None)
        return self._obs

    def step(self, action):
        self._check_step()
        a = torch.as_tensor(action, device=self.device).to(torch.int32).contiguous()
        assert a.shape == (self.num_envs,), "action must be [num_envs] ints in 0..3"
        self._launch(a)
        return self._obs, self._reward, self._done, {"steps": self.steps}


class _Maze3D(_MazeBatch):
    def __init__(self, num_envs, device, resolution, max_steps, task_type, auto_reset, collision_dist=0.20,
                 max_vision_range=12.0, fol_angle=0.6 * PI, obs_dtype=torch.int32):
        super().__init__(num_envs, device, max_steps, task_type, auto_reset)
        self.resolution_horizon, self.resolution_vertical = int(resolution[0]), int(resolution[1])
        self.collision_dist, self.max_vision_range, self.fol_angle = collision_dist, max_vision_range, fol_angle
        H, V = self.resolution_horizon, self.resolution_vertical
        self.observation_space = Box(low=np.zeros((H, V, 3), np.float32), high=np.full((H, V, 3), 256, np.float32),
                                     dtype=np.float32)      # declared like the reference (maze_env.py:38-40)
        # int32 is the reference's frame dtype (values exceed 255: ray_caster_utils.py light > 1);
        # torch.uint8 selects the saturating fast path (not reference-exact where the reference exceeds 255)
        assert obs_dtype in (torch.int32, torch.uint8)
        self.obs_dtype = obs_dtype
        self._obs = torch.zeros(self.num_envs, H, V, 3, dtype=obs_dtype, device=self.device)
        self._tex_version = None

    def _on_set_task(self):
        self._build_view()

    def _build_view(self):
        lib, dev = self._lib, self.device
        H, V = self.resolution_horizon, self.resolution_vertical
        v = _lib.MazeView()
        v.res_h, v.res_v = H, V
        v.max_vision, v.l_focal, v.text_size = self.max_vision_range, 0.20, 1.0     # maze_discrete_3d.py:113-117
        v.tan_half_fov = float(np.tan(self.fol_angle / 2))                           # ray_caster_utils.py:68
        v.collision_dist = self.collision_dist
        cc, cs = np.zeros(H, np.float64), np.zeros(H, np.float64)
        _lib.check(lib.mg_maze_view_tables(H, v.tan_half_fov, v.l_focal, cc.ctypes.data, cs.ctypes.data),
                   "mg_maze_view_tables")
        self._col_cos, self._col_sin = torch.from_numpy(cc).to(dev), torch.from_numpy(cs).to(dev)
        v.col_cos, v.col_sin = self._col_cos.data_ptr(), self._col_sin.data_ptr()
        ori = np.asarray([0.0, 0.5, 1.0, 1.5], dtype="float32") * PI                 # maze_discrete_3d.py:46
        s4, c4 = np.sin(ori).astype(np.float32), np.cos(ori).astype(np.float32)
        for i in range(4):
            v.ori_sin[i], v.ori_cos[i] = float(s4[i]), float(c4[i])
        tex, ceil = MAZE_TASK_MANAGER.packed_textures()
        self._tex_t = torch.from_numpy(tex.view(np.int32).copy()).to(dev).contiguous()   # uint32 bits
        self._ceil_t = torch.from_numpy(ceil.view(np.int32).copy()).to(dev).contiguous()
        v.textures, v.ceil_texture = self._tex_t.data_ptr(), self._ceil_t.data_ptr()
        v.n_textures, v.tex_size = tex.shape[0], tex.shape[1]
        v.obs_format = 1 if self.obs_dtype == torch.uint8 else 0
        # DDA_2D (ray_caster_utils.py:31) stops once hit_dist >= max_vision: a ray crosses at most
        # floor(max_vision / cell_size) + 1 cell boundaries per axis plus the final overshoot, so it can
        # record at most 2*floor(mv/cs) + 4 translucent cells (incl. the start cell) whatever n is
        min_cs = self._min_cell_size
        v.max_ray_records = 2 * int(self.max_vision_range / min_cs) + 5
        v.uniform_cell_size = self._uniform_cell_size
        self._view_c = v
        self._tex_version = MAZE_TASK_MANAGER.version

    def _launch(self, action, continuous):
        if self._tex_version != MAZE_TASK_MANAGER.version:
            self._build_view()
        rc = self._lib.mg_maze3d_step(self._tasks_c, self._view_c, self._tt, self.max_steps, int(continuous),
                                      int(self.auto_reset), self.num_envs, self._state_c, _lib.ptr(action),
                                      _lib.ptr(self._obs), _lib.ptr(self._reward), _lib.ptr(self._reward64),
                                      _lib.ptr(self._done), _lib.current_stream(self.device))
        _lib.check(rc, "mg_maze3d_step")


class MetaMazeDiscrete3D(_Maze3D):
    """maze_env.py:16-83. action int in {0..3}: turn left / right, step back / forward."""

    def __init__(self, num_envs=1, device="cuda", enable_render=False, render_scale=480, resolution=(320, 320),
                 max_steps=5000, task_type="SURVIVAL", auto_reset=False, obs_dtype=torch.int32):
        super().__init__(num_envs, device, resolution, max_steps, task_type, auto_reset, obs_dtype=obs_dtype)
        self.action_space = Discrete(4)

    def _observe(self):
        self._launch(None, False)
        return self._obs

    def step(self, action):
        self._check_step()
        a = torch.as_tensor(action, device=self.device).to(torch.int32).contiguous()
        assert a.shape == (self.num_envs,), "action must be [num_envs] ints in 0..3"
        self._launch(a, False)
        return self._obs, self._reward, self._done, {"steps": self.steps}


class MetaMazeContinuous3D(_Maze3D):
    """maze_env.py:85-153. action float32 [N,2] = (turn rate, walk speed), each clipped to [-1,1]."""

    def __init__(self, num_envs=1, device="cuda", enable_render=False, render_scale=480, resolution=(320, 320),
                 max_steps=5000, task_type="SURVIVAL", auto_reset=False, obs_dtype=torch.int32):
        super().__init__(num_envs, device, resolution, max_steps, task_type, auto_reset, obs_dtype=obs_dtype)
        self.action_space = Box(low=np.array([-1.0, -1.0]), high=np.array([1.0, 1.0]), dtype=np.float32)

    def _observe(self):
        self._launch(None, True)
        return self._obs

    def step(self, action):
        self._check_step()
        a = torch.as_tensor(action, dtype=torch.float32, device=self.device).contiguous()
        assert a.shape == (self.num_envs, 2), "action must be [num_envs, 2]"
        self._launch(a, True)
        return self._obs, self._reward, self._done, {"steps": self.steps}
