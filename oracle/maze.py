"""ctypes front-end of oracle/maze_oracle.c — TEST INFRASTRUCTURE (see oracle/__init__.py)."""
import ctypes as C

import numpy as np

from .quadrotor import build, _LIB_PATH  # same shared object, same Makefile

ESCAPE, SURVIVAL = 0, 1
TASK_TYPES = {"ESCAPE": ESCAPE, "SURVIVAL": SURVIVAL}
PI = 3.1415926  # dynamics.py:6


class CTask(C.Structure):
    _fields_ = [("n", C.c_int32), ("start", C.c_int32 * 2), ("goal", C.c_int32 * 2),
                ("walls", C.c_void_p), ("texts", C.c_void_p), ("food_rewards", C.c_void_p),
                ("food_interval", C.c_void_p),
                ("cell_size", C.c_double), ("wall_height", C.c_double), ("agent_height", C.c_double),
                ("initial_life", C.c_double), ("max_life", C.c_double), ("step_reward", C.c_double),
                ("goal_reward", C.c_double)]


class CState(C.Structure):
    _fields_ = [("grid", C.c_int32 * 2), ("steps", C.c_int32), ("ori_idx", C.c_int32), ("ori", C.c_double),
                ("loc", C.c_float * 2), ("life", C.c_double), ("cur_food", C.c_void_p),
                ("wait_refresh", C.c_void_p), ("revival", C.c_void_p)]


class CView(C.Structure):
    _fields_ = [("H", C.c_int32), ("V", C.c_int32), ("max_vision", C.c_double), ("l_focal", C.c_double),
                ("text_size", C.c_double), ("tan_half_fov", C.c_double), ("textures", C.c_void_p),
                ("ceil_tex", C.c_void_p), ("tex_size", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.mo_evaluation_rule.restype = C.c_int
        _lib.mo_step_2d.restype = C.c_int
        _lib.mo_step_disc3d.restype = C.c_int
        _lib.mo_step_cont3d.restype = C.c_int
        _lib.mo_step_cont3d.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_double,
                                        C.c_double, C.c_void_p]
    return _lib


def discrete_ori_tables():
    """The 4 headings of MazeCoreDiscrete3D (maze_discrete_3d.py:46) and their float32 sin/cos,
    computed exactly as the reference does (numpy float32 ufuncs)."""
    ori = np.asarray([0.0, 0.5, 1.0, 1.5], dtype="float32") * PI
    return np.sin(ori).astype(np.float32), np.cos(ori).astype(np.float32)


class Task(object):
    """Owns the numpy arrays behind a CTask. Fields as in TaskConfig (maze_task.py:15-17)."""

    def __init__(self, start, goal, cell_walls, cell_texts, cell_size, wall_height, agent_height, initial_life,
                 max_life, step_reward, goal_reward, food_rewards, food_interval):
        self.walls = np.ascontiguousarray(cell_walls, np.int32)
        self.texts = np.ascontiguousarray(cell_texts, np.int32)
        self.food_rewards = np.ascontiguousarray(food_rewards, np.float64)
        self.food_interval = np.ascontiguousarray(food_interval, np.int32)
        self.n = self.walls.shape[0]
        c = CTask()
        c.n = self.n
        c.start[0], c.start[1] = int(start[0]), int(start[1])
        c.goal[0], c.goal[1] = int(goal[0]), int(goal[1])
        c.walls = self.walls.ctypes.data
        c.texts = self.texts.ctypes.data
        c.food_rewards = self.food_rewards.ctypes.data
        c.food_interval = self.food_interval.ctypes.data
        c.cell_size, c.wall_height, c.agent_height = float(cell_size), float(wall_height), float(agent_height)
        c.initial_life, c.max_life = float(initial_life), float(max_life)
        c.step_reward, c.goal_reward = float(step_reward), float(goal_reward)
        self.c = c

    @classmethod
    def from_golden(cls, g):
        return cls(**{k[5:]: g[k] for k in g.files if k.startswith("task_")})


class State(object):
    def __init__(self, task):
        nn = task.n * task.n
        self.cur_food = np.zeros(nn, np.float64)
        self.wait = np.zeros(nn, np.int32)
        self.revival = np.zeros(nn, np.int32)
        c = CState()
        c.cur_food, c.wait_refresh, c.revival = self.cur_food.ctypes.data, self.wait.ctypes.data, self.revival.ctypes.data
        self.c = c


class View(object):
    """Renderer constants. fov = 0.6*PI, l_focal 0.20, max_vision 12.0, text_size 1.0
    (maze_discrete_3d.py:21-23,113-117)."""

    def __init__(self, textures_u8, ceil_u8, H, V, max_vision=12.0, l_focal=0.20, text_size=1.0, fov=0.6 * PI):
        self.tex = np.ascontiguousarray(textures_u8, np.float32)
        self.ceil = np.ascontiguousarray(ceil_u8, np.uint8)
        c = CView()
        c.H, c.V = int(H), int(V)
        c.max_vision, c.l_focal, c.text_size = max_vision, l_focal, text_size
        c.tan_half_fov = float(np.tan(fov / 2))       # numpy.tan like ray_caster_utils.py:68
        c.textures, c.ceil_tex = self.tex.ctypes.data, self.ceil.ctypes.data
        c.tex_size = self.tex.shape[1]
        self.c = c
        self.H, self.V = int(H), int(V)
        self.sin4, self.cos4 = discrete_ori_tables()


def reset(task, task_type, st):
    lib().mo_reset(C.byref(task.c), C.c_int(task_type), C.byref(st.c))


def step_2d(task, task_type, max_steps, st, action):
    r = C.c_double()
    d = lib().mo_step_2d(C.byref(task.c), task_type, max_steps, C.byref(st.c), int(action), C.byref(r))
    return r.value, bool(d)


def observe_2d(task, task_type, st, view_grid):
    w = 2 * view_grid + 1
    obs = np.zeros((w, w), np.float32)
    lib().mo_observe_2d(C.byref(task.c), task_type, C.byref(st.c), view_grid, obs.ctypes.data_as(C.c_void_p))
    return obs


def step_disc3d(task, task_type, max_steps, st, action):
    r = C.c_double()
    d = lib().mo_step_disc3d(C.byref(task.c), task_type, max_steps, C.byref(st.c), int(action), C.byref(r))
    return r.value, bool(d)


def step_cont3d(task, task_type, max_steps, st, turn, walk, collision_dist=0.20):
    r = C.c_double()
    d = lib().mo_step_cont3d(C.byref(task.c), task_type, max_steps, collision_dist, C.byref(st.c), float(turn),
                             float(walk), C.byref(r))
    return r.value, bool(d)


def observe_3d(task, task_type, view, st, continuous):
    rgb = np.zeros((view.H, view.V, 3), np.int32)
    lib().mo_observe_3d(C.byref(task.c), task_type, C.byref(view.c), C.byref(st.c), int(continuous),
                        view.sin4.ctypes.data_as(C.c_void_p), view.cos4.ctypes.data_as(C.c_void_p),
                        rgb.ctypes.data_as(C.c_void_p))
    return rgb
