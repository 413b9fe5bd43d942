"""oracle/walker_oracle.c (the native CPU baseline of bench.py's C4 entry and the source of the counted flop figure) against
the numpy restatement oracle/abd.py: same physics, different formulation (composite-rigid-body + whitened solver vs Jacobian
form + Delassus matrix) — states to 1e-12 per sub-step, whole env steps (observation, reward terms, done) over trajectories
with landings, joint-limit pushes and self-collision. CPU only."""
import ctypes as C

import numpy as np
import pytest

from oracle import abd, walker_c
from walker_fixtures import load_models, world_kw

MODELS = load_models()


def _c_env(m, ant):
    lib = walker_c.load()
    power = np.full(len(m.joint_lo), 100.0) * 2.5 if ant else abd.HUMANOID_MOTOR_POWER * 0.41
    cm, table = walker_c.make_model(m, power)
    prm = walker_c.ant_params(m) if ant else walker_c.humanoid_params(m)
    return lib, cm, table, prm, walker_c.Env()


def _np_env(m, ant, **kw):
    if ant:
        return abd.WalkerEnv(m, prm=abd.Params(friction=0.8 * float(m.geom_friction), power=2.5, self_friction=float(m.geom_friction) ** 2, **world_kw(m)),
                             motor_power=np.full(len(m.joint_lo), 100.0), alive_z=0.26, alive_bonus=1.0, initial_z=None, torque_f32=False, **kw)
    return abd.WalkerEnv(m, prm=abd.Params(friction=0.8 * float(m.geom_friction), self_friction=float(m.geom_friction) ** 2, **world_kw(m)), **kw)


def _state_err(cs, s, nj):
    return max(np.abs(np.array(cs.pos[:]) - s.pos).max(), np.abs(np.array(cs.rot[:]).reshape(3, 3) - s.rot).max(),
               np.abs(np.array(cs.vel[:]) - s.v).max(), np.abs(np.array(cs.omega[:]) - s.w).max(),
               np.abs(np.array(cs.q[:nj]) - s.q).max(), np.abs(np.array(cs.qd[:nj]) - s.qd).max())


@pytest.mark.parametrize("name", ["humanoid", "humanoid_tra_137", "ant", "ant_tra_005", "humanoid@mujoco", "ant@mujoco"])
def test_c_env_step_matches_numpy_trajectory(name):
    m = MODELS[name]
    ant = name.startswith("ant")
    nj = len(m.joint_lo)
    lib, cm, table, prm, env = _c_env(m, ant)
    o = _np_env(m, ant)
    rs = np.random.RandomState(7)
    noise = rs.uniform(-0.1, 0.1, nj)
    obs_c = np.zeros(8 + 2 * nj + len(m.foot_body), np.float32)
    lib.wo_env_reset(C.byref(cm), C.byref(prm), C.byref(env), noise.ctypes.data_as(C.POINTER(C.c_double)), obs_c.ctypes.data_as(C.POINTER(C.c_float)))
    obs_n = o.reset(noise)
    assert np.array_equal(obs_c, obs_n)
    worst = 0.0
    rew, r5 = C.c_double(), (C.c_double * 5)()
    for t in range(30):          # 120 sub-steps: the fall, the landing, joint limits, thighs / arms touching
        a = rs.uniform(-1.3, 1.3, nj).astype(np.float32)
        done_c = lib.wo_env_step(C.byref(cm), C.byref(prm), C.byref(env), a.ctypes.data_as(C.POINTER(C.c_float)),
                                 obs_c.ctypes.data_as(C.POINTER(C.c_float)), C.byref(rew), r5)
        obs_n, rew_n, done_n, info = o.step(a)
        worst = max(worst, _state_err(env.s, o.s, nj))
        assert worst < 1e-7, (t, worst)                      # free-running: round-off grows through the contacts (one sub-step: 1e-12, below)
        assert np.allclose(obs_c, obs_n, rtol=0, atol=2e-6) and bool(done_c) == bool(done_n)
        assert np.allclose(list(r5), info["rewards"], rtol=0, atol=1e-6) and abs(rew.value - rew_n) < 1e-6
        assert np.array_equal(np.array(env.feet_contact[:len(m.foot_body)]), o.feet_contact)


@pytest.mark.parametrize("name", ["humanoid", "ant", "humanoid@mujoco", "ant@mujoco"])
def test_c_substep_matches_numpy_one_substep_at_a_time(name):
    """Every sub-step from the numpy engine's own state: 1e-12 (no accumulation), with contacts, limits and self-collision rows."""
    m = MODELS[name]
    ant = name.startswith("ant")
    nj = len(m.joint_lo)
    lib, cm, table, prm, _ = _c_env(m, ant)
    o = _np_env(m, ant)
    rs = np.random.RandomState(11)
    o.reset(rs.uniform(-0.1, 0.1, nj))
    worst, rows_seen = 0.0, 0
    for t in range(160):
        if t % 4 == 0:
            tau = o.torques(rs.uniform(-1.3, 1.3, nj).astype(np.float32))
        cs = walker_c.State()
        cs.pos[:], cs.rot[:], cs.vel[:], cs.omega[:] = list(o.s.pos), list(o.s.rot.reshape(9)), list(o.s.v), list(o.s.w)
        cs.q[:nj], cs.qd[:nj] = list(o.s.q), list(o.s.qd)
        touch = (C.c_ulonglong * 2)()
        nr = lib.wo_substep(C.byref(cm), C.byref(prm), C.byref(cs), np.ascontiguousarray(tau).ctypes.data_as(C.POINTER(C.c_double)), touch)
        touching = abd.substep(m, o.s, tau, o.prm)
        rows_seen += nr
        worst = max(worst, _state_err(cs, o.s, nj))
        assert worst < 1e-12 * max(1.0, np.abs(o.s.u()).max()), (t, worst)
        assert {g for g in range(len(m.sph_body)) if (touch[g >> 6] >> (g & 63)) & 1} == set(touching)
    assert rows_seen > 300
