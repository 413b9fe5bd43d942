"""ctypes front-end of oracle/quadrotor_oracle.c — TEST INFRASTRUCTURE (see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmetagym_oracle.so")

TASK_NO_COLLISION = 0
TASK_VELOCITY = 1
TASK_HOVERING = 2


def build(force=False):
    """Compile the oracle with gcc (seconds). Called by __graft_entry__.build() and on first use."""
    srcs = [f for f in os.listdir(_HERE) if f.endswith(("_oracle.c", ".h")) or f == "Makefile"]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH) for f in srcs)
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _LIB_PATH


class Consts(C.Structure):
    _fields_ = [
        ("precision", C.c_double), ("quality", C.c_double),
        ("ct0", C.c_double), ("ct1", C.c_double), ("ct2", C.c_double),
        ("mm", C.c_double), ("jm", C.c_double), ("ra", C.c_double), ("phi", C.c_double),
        ("fail_velocity", C.c_double), ("fail_w", C.c_double), ("fail_range", C.c_double),
        ("min_voltage", C.c_double), ("max_voltage", C.c_double),
        ("dt", C.c_double), ("healthy_reward", C.c_double), ("z_offset", C.c_double),
        ("x_offset", C.c_int64), ("y_offset", C.c_int64),
        ("nt", C.c_int32), ("task", C.c_int32),
        ("inertia_inv", C.c_float * 9), ("drag_m", C.c_float * 9), ("drag_f", C.c_float * 9),
        ("gravity_center", C.c_float * 3), ("prop_coord", C.c_float * 12),
        ("map", C.POINTER(C.c_int32)), ("map_h", C.c_int32), ("map_w", C.c_int32),
        ("velocity_targets", C.POINTER(C.c_float)),
    ]


class State(C.Structure):
    _fields_ = [
        ("pos", C.c_float * 3), ("vel", C.c_double * 3), ("omega", C.c_double * 3),
        ("propw", C.c_float * 4), ("R", C.c_float * 9), ("Rinv", C.c_float * 9),
        ("power", C.c_float), ("pos0_z", C.c_float),
    ]


class AutoReset(C.Structure):
    """qo_autoreset — same fields as the product's mg_quadrotor_autoreset"""
    _fields_ = [("init_velocity", C.c_float * 3), ("init_angular_velocity", C.c_float * 3),
                ("init_velocity_noisy", C.c_double), ("init_angular_velocity_noisy", C.c_double),
                ("seed", C.c_uint64), ("env_id_base", C.c_uint64)]


def default_autoreset(seed=0, env_id_base=0):
    """config.json: init_velocity / init_angular_velocity = 0 +- noisy 2.0 / 5.0"""
    ar = AutoReset()
    ar.init_velocity_noisy, ar.init_angular_velocity_noisy = 2.0, 5.0
    ar.seed, ar.env_id_base = seed, env_id_base
    return ar


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.qo_sizeof_state.restype = C.c_size_t
        _lib.qo_sizeof_consts.restype = C.c_size_t
        assert _lib.qo_sizeof_state() == C.sizeof(State), "oracle State layout drifted"
        assert _lib.qo_sizeof_consts() == C.sizeof(Consts), "oracle Consts layout drifted"
    return _lib


def default_consts(nt=1000, task=TASK_HOVERING):
    c = Consts()
    lib().qo_default_consts(C.byref(c))
    c.nt = nt
    c.task = task
    return c


def consts_from_config(cfg, nt=1000, task=TASK_HOVERING, dt=0.01):
    """`_parse_cfg` (quadrotorsim.py:50-109) for a config dict with the config.json schema: scalars stay
    python floats (doubles), the matrices are float32, inverse inertia = inv of the float32 matrix."""
    c = default_consts(nt=nt, task=task)
    c.precision, c.quality, c.dt = float(cfg["precision"]), float(cfg["quality"]), float(dt)
    i = cfg["inertia"]
    inertia = np.array([[i["xx"], i["xy"], i["xz"]], [i["xy"], i["yy"], i["yz"]], [i["xz"], i["yz"], i["zz"]]],
                       np.float32)
    c.inertia_inv[:] = inv3(inertia).reshape(9).tolist()
    d = cfg["drag"]
    for k in range(9):
        c.drag_m[k] = c.drag_f[k] = 0.0
    c.drag_m[0], c.drag_m[4], c.drag_m[8] = float(d["m_xx"]), float(d["m_yy"]), float(d["m_zz"])
    c.drag_f[0], c.drag_f[4], c.drag_f[8] = float(d["f_xx"]), float(d["f_yy"]), float(d["f_zz"])
    g = cfg["gravity_center"]
    c.gravity_center[:] = [float(g["x"]), float(g["y"]), float(g["z"])]
    t = cfg["thrust"]
    c.ct0, c.ct1, c.ct2 = float(t["CT"][0]), float(t["CT"][1]), float(t["CT"][2])
    c.mm, c.jm, c.phi, c.ra = float(t["Mm"]), float(t["Jm"]), float(t["phi"]), float(t["RA"])
    f = cfg["fail"]
    c.fail_velocity, c.fail_range, c.fail_w = float(f["velocity"]), float(f["range"]), float(f["w"])
    c.prop_coord[:] = [float(cfg["propeller"][p][ax]) for p in range(4) for ax in "xyz"]
    c.max_voltage, c.min_voltage = float(cfg["electric"]["max_voltage"]), float(cfg["electric"]["min_voltage"])
    return c


def make_states(pos, vel, omega, propw, R):
    """Arrays [n,3],[n,3],[n,3],[n,4],[n,9] -> ctypes array of State with Rinv = inv(R)."""
    n = len(pos)
    arr = (State * n)()
    L = lib()
    for e in range(n):
        s = arr[e]
        s.pos[:] = [float(x) for x in np.asarray(pos[e], np.float32)]
        s.vel[:] = [float(x) for x in np.asarray(vel[e], np.float64)]
        s.omega[:] = [float(x) for x in np.asarray(omega[e], np.float64)]
        s.propw[:] = [float(x) for x in np.asarray(propw[e], np.float32)]
        s.R[:] = [float(x) for x in np.asarray(R[e], np.float32).reshape(9)]
        s.power = 0.0
        s.pos0_z = 0.0
        L.qo_refresh_inverse(C.byref(s))
    return arr


def states_to_arrays(arr):
    n = len(arr)
    out = dict(pos=np.zeros((n, 3), np.float32), vel=np.zeros((n, 3), np.float64),
               omega=np.zeros((n, 3), np.float64), propw=np.zeros((n, 4), np.float32),
               R=np.zeros((n, 9), np.float32), power=np.zeros(n, np.float32))
    for e in range(n):
        s = arr[e]
        out["pos"][e] = s.pos[:]
        out["vel"][e] = s.vel[:]
        out["omega"][e] = s.omega[:]
        out["propw"][e] = s.propw[:]
        out["R"][e] = s.R[:]
        out["power"][e] = s.power
    return out


def batch_env_step(consts, states, ct, actions):
    """One env.step for every env. ct: int32[n] (updated in place). actions: f32[n,4].
    Returns obs f32[n,16], reward f64[n], done i32[n], failed i32[n]."""
    n = len(states)
    actions = np.ascontiguousarray(actions, np.float32)
    assert actions.shape == (n, 4) and ct.dtype == np.int32
    obs = np.zeros((n, 16), np.float32)
    reward = np.zeros(n, np.float64)
    done = np.zeros(n, np.int32)
    failed = np.zeros(n, np.int32)
    lib().qo_batch_env_step(C.byref(consts), C.c_int(n), states, ct.ctypes.data_as(C.c_void_p),
                            actions.ctypes.data_as(C.c_void_p), obs.ctypes.data_as(C.c_void_p),
                            reward.ctypes.data_as(C.c_void_p), done.ctypes.data_as(C.c_void_p),
                            failed.ctypes.data_as(C.c_void_p))
    return obs, reward, done, failed


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*[int(x) & 0xFFFFFFFF for x in ctr])
    k = (C.c_uint32 * 2)(*[int(x) & 0xFFFFFFFF for x in key])
    out = (C.c_uint32 * 4)()
    lib().qo_philox4x32_10(c, k, out)
    return [int(x) for x in out]


def reset_noise(ar, gid, episode):
    """(velocity[3], body rate[3]) of the `episode`-th auto-reset of global env `gid`."""
    v, w = (C.c_double * 3)(), (C.c_double * 3)()
    lib().qo_reset_noise(C.byref(ar), C.c_uint64(int(gid)), C.c_uint32(int(episode)), v, w)
    return np.array(v[:]), np.array(w[:])


def batch_env_step_autoreset(consts, ar, states, ct, episode, actions):
    """batch_env_step with the fused reset. episode: uint32[n], updated in place."""
    n = len(states)
    actions = np.ascontiguousarray(actions, np.float32)
    assert actions.shape == (n, 4) and ct.dtype == np.int32 and episode.dtype == np.uint32
    obs = np.zeros((n, 16), np.float32)
    reward = np.zeros(n, np.float64)
    done = np.zeros(n, np.int32)
    failed = np.zeros(n, np.int32)
    lib().qo_batch_env_step_autoreset(C.byref(consts), C.byref(ar), C.c_int(n), states, ct.ctypes.data_as(C.c_void_p),
                                      episode.ctypes.data_as(C.c_void_p), actions.ctypes.data_as(C.c_void_p),
                                      obs.ctypes.data_as(C.c_void_p), reward.ctypes.data_as(C.c_void_p),
                                      done.ctypes.data_as(C.c_void_p), failed.ctypes.data_as(C.c_void_p))
    return obs, reward, done, failed


def observe(consts, states):
    n = len(states)
    obs = np.zeros((n, 16), np.float32)
    L = lib()
    for e in range(n):
        L.qo_observe(C.byref(consts), C.byref(states[e]), obs[e].ctypes.data_as(C.c_void_p))
    return obs


def inv3(A):
    A = np.ascontiguousarray(A, np.float32).reshape(9)
    out = np.zeros(9, np.float32)
    lib().qo_inv3_f32(A.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out.reshape(3, 3)


def batch_run(consts, states, init_states, ct, actions, iters):
    """iters env-steps per env inside one C call (actions f32 [B][n][4] cycled); finished episodes
    restart from init_states. Returns the number of env-steps done."""
    n = len(states)
    actions = np.ascontiguousarray(actions, np.float32)
    assert actions.ndim == 3 and actions.shape[1:] == (n, 4) and len(init_states) == n
    fn = lib().qo_batch_run
    fn.restype = C.c_long
    return fn(C.byref(consts), C.c_int(n), states, init_states, ct.ctypes.data_as(C.c_void_p),
              actions.ctypes.data_as(C.c_void_p), C.c_int(actions.shape[0]), C.c_int(iters))


def velocity_target_actions(seed, nt, lo=0.10, hi=15.0):
    """The action stream define_velocity_control_task draws (quadrotorsim.py:306-315):
    np.random.seed(seed), then nt calls of uniform(low, high, size=4).astype(float32)."""
    rs = np.random.RandomState(seed)
    return np.stack([rs.uniform(low=lo, high=hi, size=4).astype(np.float32) for _ in range(nt)])


def velocity_targets(consts, actions):
    actions = np.ascontiguousarray(actions, np.float32)
    out = np.zeros((len(actions), 3), np.float32)
    lib().qo_velocity_targets(C.byref(consts), C.c_int(len(actions)), actions.ctypes.data_as(C.c_void_p),
                              out.ctypes.data_as(C.c_void_p))
    return out


def env_step_velocity(consts, state, ct, action):
    """One velocity_control env.step for ONE env (state: State, ct: c_int). Returns obs[19], reward, done, failed."""
    obs = np.zeros(19, np.float32)
    r, d = C.c_double(), C.c_int()
    a = np.ascontiguousarray(action, np.float32)
    f = lib().qo_env_step_velocity(C.byref(consts), C.byref(state), C.byref(ct), a.ctypes.data_as(C.c_void_p),
                                   obs.ctypes.data_as(C.c_void_p), C.byref(r), C.byref(d))
    return obs, r.value, bool(d.value), int(f)
