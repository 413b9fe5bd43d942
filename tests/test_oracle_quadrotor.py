"""Pin oracle/quadrotor_oracle.c against the golden vectors produced by the unmodified reference
(oracle/gen_golden.py -> tests/golden/quadrotor_*.npz). CPU-only."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import quadrotor as qo
from parity import REL_TOL, obs_rel_err, scalar_rel_err, vec_rel_err


def consts_for(path, g):
    """Golden files named *_custom_* were generated with tests/golden/quadrotor_custom_config.json."""
    if "_custom_" in os.path.basename(path):
        with open(os.path.join(os.path.dirname(path), "quadrotor_custom_config.json")) as f:
            return qo.consts_from_config(json.load(f), nt=int(g["nt"]))
    return qo.default_consts(nt=int(g["nt"]))


def _rollout(g, path=""):
    c = consts_for(path, g)
    st = qo.make_states(g["init_pos"][None], g["init_vel"][None], g["init_omega"][None],
                        g["init_propw"][None], g["init_R"][None])
    ct = np.zeros(1, np.int32)
    T = len(g["reward"])
    rec = {k: [] for k in ("pos", "vel", "omega", "propw", "R", "power", "obs", "reward", "done", "ct")}
    for t in range(T):
        obs, rew, done, failed = qo.batch_env_step(c, st, ct, g["actions"][t][None])
        assert failed[0] == 0
        s = qo.states_to_arrays(st)
        for k in ("pos", "vel", "omega", "propw", "R", "power"):
            rec[k].append(s[k][0])
        rec["obs"].append(obs[0])
        rec["reward"].append(rew[0])
        rec["done"].append(bool(done[0]))
        rec["ct"].append(int(ct[0]))
    return {k: np.asarray(v) for k, v in rec.items()}, c, st


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden",
                                                                "quadrotor_traj_*.npz"))))
def test_oracle_matches_reference_rollout(path):
    g = np.load(path)
    assert str(g["numpy_version"]).startswith("2."), "goldens must come from the NEP-50 numpy"
    rec, c, _ = _rollout(g, path)
    # reset observation
    st0 = qo.make_states(g["init_pos"][None], g["init_vel"][None], g["init_omega"][None],
                         g["init_propw"][None], g["init_R"][None])
    assert obs_rel_err(qo.observe(c, st0), g["obs0"][None]) < REL_TOL
    # integer/boolean outputs are exact
    assert np.array_equal(rec["done"], g["done"])
    assert np.array_equal(rec["ct"], g["ct"])
    # floating-point trajectory: whole rollout (400-1000 env steps = 4-10k Euler sub-steps)
    # the simulator state is BIT-IDENTICAL to the reference over the whole rollout
    for k in ("pos", "vel", "omega", "propw", "R", "power"):
        assert np.array_equal(rec[k], g[k]), "state %s is not bit-identical to the reference" % k
    # ... and so are the reward and every observation entry except the three atan2f angles (libm)
    assert np.array_equal(rec["reward"], g["reward"])
    nonang = [i for i in range(16) if i not in (12, 13, 14)]
    assert np.array_equal(rec["obs"][:, nonang], g["obs"][:, nonang])
    errs = dict(
        pos=vec_rel_err(rec["pos"], g["pos"]), vel=vec_rel_err(rec["vel"], g["vel"]),
        omega=vec_rel_err(rec["omega"], g["omega"]), propw=vec_rel_err(rec["propw"], g["propw"]),
        R=vec_rel_err(rec["R"], g["R"]), obs=obs_rel_err(rec["obs"], g["obs"]),
        reward=scalar_rel_err(rec["reward"], g["reward"]), power=scalar_rel_err(rec["power"], g["power"]))
    print(os.path.basename(path), {k: "%.2e" % v for k, v in errs.items()})
    for k, v in errs.items():
        assert v < REL_TOL, (k, v)


def test_oracle_matches_reference_single_steps():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "quadrotor_onestep.npz"))
    c = qo.default_consts()
    st = qo.make_states(g["in_pos"], g["in_vel"], g["in_omega"], g["in_propw"], g["in_R"])
    ct = np.zeros(len(st), np.int32)
    obs, rew, done, failed = qo.batch_env_step(c, st, ct, g["actions"])
    s = qo.states_to_arrays(st)
    assert not failed.any()
    assert np.array_equal(done.astype(bool), g["done"])
    # pure-f32 recurrences with no BLAS/LAPACK in the loop are bit-exact
    assert np.array_equal(s["propw"], g["out_propw"])
    assert np.array_equal(s["power"], g["power"])
    for k in ("pos", "vel", "omega", "R"):
        assert np.array_equal(s[k], g["out_" + k]), k
    assert obs_rel_err(obs, g["obs"]) < 1e-6
    assert scalar_rel_err(rew, g["reward"]) < 1e-6


def test_oracle_failure_flags():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "quadrotor_fail.npz"))
    c = qo.default_consts()
    st = qo.make_states(g["in_pos"], g["in_vel"], g["in_omega"], g["in_propw"], g["in_R"])
    ct = np.zeros(len(st), np.int32)
    obs, rew, done, failed = qo.batch_env_step(c, st, ct, g["actions"])
    assert np.array_equal(failed != 0, g["failed"])
    # the state at the raise (sub-step applied, then frozen) matches the reference's
    s = qo.states_to_arrays(st)
    for k in ("pos", "vel", "omega", "propw", "R"):
        assert vec_rel_err(s[k], g["out_" + k]) < REL_TOL, k
    assert done[g["failed"]].all()


def test_inverse_close_to_lapack():
    """f64-adjugate inverse vs np.linalg.inv on drifted rotation matrices like the ones the path sees
    (R = rotation * (1 + 2e-3 noise))."""
    rs = np.random.RandomState(0)
    worst = 0.0
    for _ in range(500):
        q = rs.normal(size=4)
        q /= np.linalg.norm(q)
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        A = (R * (1 + rs.uniform(-2e-3, 2e-3, (3, 3)))).astype(np.float32)
        worst = max(worst, vec_rel_err(qo.inv3(A).reshape(1, 9), np.linalg.inv(A).reshape(1, 9)))
    assert worst == 0.0, worst      # correctly rounded, like numpy's dgesv-then-cast


def test_oracle_no_collision_on_reference_map():
    """task='no_collision' with the reference's obstacle map (golden recorded from the reference):
    done / reward / position sequences are reproduced exactly."""
    import ctypes as C
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "quadrotor_no_collision_map.npz"))
    grid = np.array(g["map"], np.int32)
    ys, xs = np.where(grid == -1)
    grid[ys[0], xs[0]] = 0
    grid = np.ascontiguousarray(grid)
    for k in range(3):
        c = qo.default_consts(nt=int(g["nt"]), task=qo.TASK_NO_COLLISION)
        c.map = grid.ctypes.data_as(C.POINTER(C.c_int32))
        c.map_h, c.map_w = grid.shape
        c.x_offset, c.y_offset = int(xs[0]), int(ys[0])
        st = qo.make_states(np.zeros((1, 3), np.float32), g["init_vel_%d" % k][None], g["init_omega_%d" % k][None],
                            np.zeros((1, 4), np.float32), np.eye(3, dtype=np.float32).reshape(1, 9))
        ct = np.zeros(1, np.int32)
        acts = g["actions_%d" % k]
        for t in range(len(acts)):
            obs, rew, done, failed = qo.batch_env_step(c, st, ct, acts[t][None])
            assert bool(done[0]) == bool(g["done_%d" % k][t]), (k, t)
            assert rew[0] == g["reward_%d" % k][t], (k, t)
            assert np.array_equal(qo.states_to_arrays(st)["pos"][0], g["pos_%d" % k][t]), (k, t)
            assert obs_rel_err(obs, g["obs_%d" % k][t][None]) < REL_TOL


def test_oracle_velocity_control_matches_reference():
    """task='velocity_control': the target trajectory (rolled by the reference at __init__ in its
    all-float32 pre-reset state, quadrotorsim.py:306-319) and a 260-step rollout across ct == nt."""
    import ctypes as C
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "quadrotor_velocity_control.npz"))
    nt = int(g["nt"])
    c = qo.default_consts(nt=nt, task=qo.TASK_VELOCITY)
    c.x_offset = c.y_offset = 0
    c.z_offset = 0.0
    tg = qo.velocity_targets(c, qo.velocity_target_actions(int(g["seed"]), nt))
    assert np.array_equal(tg, g["targets"])                  # the all-float32 choreography, bit for bit
    c.velocity_targets = tg.ctypes.data_as(C.POINTER(C.c_float))
    st = qo.make_states(np.zeros((1, 3), np.float32), g["init_vel"][None], g["init_omega"][None],
                        np.zeros((1, 4), np.float32), np.eye(3, dtype=np.float32).reshape(1, 9))
    ct = C.c_int(0)
    for t in range(len(g["actions"])):
        obs, r, d, f = qo.env_step_velocity(c, st[0], ct, g["actions"][t])
        assert f == 0 and d == bool(g["done"][t]) and ct.value == g["ct"][t], t
        assert obs_rel_err(obs[None, :16], g["obs"][t][None, :16], z_offset=1.0) < REL_TOL, t
        nonang = [i for i in range(19) if i not in (12, 13, 14)]
        assert np.array_equal(obs[nonang], g["obs"][t][nonang]), t
        assert r == g["reward"][t], t
    assert g["done"].sum() == 1


# ---- fused auto-reset restatement (the checker of the launch bench.py times) -------------------------------

RANDOM123_KAT = [   # Random123 kat_vectors, philox4x32 10: counter[4], key[2] -> output[4]
    ([0x00000000] * 4, [0x00000000] * 2, [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


def _philox_python(ctr, key):
    """Philox4x32-10 from the paper's definition (Salmon et al., SC'11, section 3.3 / Random123 philox.h), written
    independently of the C restatement: S-box = mulhilo by two constants, key bumped by Weyl constants per round."""
    M0, M1, W0, W1, MASK = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF
    c, k = [int(x) for x in ctr], [int(x) for x in key]
    for r in range(10):
        hi0, lo0 = divmod(M0 * c[0], 1 << 32)
        hi1, lo1 = divmod(M1 * c[2], 1 << 32)
        c = [hi1 ^ c[1] ^ k[0], lo1, hi0 ^ c[3] ^ k[1], lo0]
        k = [(k[0] + W0) & MASK, (k[1] + W1) & MASK]
    return c


def test_philox_known_answers():
    for ctr, key, want in RANDOM123_KAT:
        assert qo.philox4x32_10(ctr, key) == want
        assert _philox_python(ctr, key) == want
    rs = np.random.RandomState(0)
    for _ in range(200):
        v = [int(x) for x in rs.randint(0, 2 ** 32, 6, dtype=np.uint64)]
        assert qo.philox4x32_10(v[:4], v[4:]) == _philox_python(v[:4], v[4:])


def test_reset_noise_formula_and_stream_separation():
    """base + noisy * U[0,1) * (+-1) per component (quadrotorsim.py:242-254) from the two Philox blocks keyed by
    (seed; global env id, episode); distinct envs / episodes / seeds get distinct draws; statistics are uniform."""
    seed = (0xABCD << 32) | 17
    ar = qo.default_autoreset(seed=seed)
    ar.init_velocity[:] = [0.5, -0.25, 0.125]
    ar.init_angular_velocity[:] = [1.0, 2.0, -3.0]
    for gid, ep in ((0, 0), (5, 3), ((1 << 40) + 9, 70000)):
        w = []
        for d in range(2):
            w += _philox_python([gid & 0xFFFFFFFF, gid >> 32, ep, d], [seed & 0xFFFFFFFF, seed >> 32])
        v, om = qo.reset_noise(ar, gid, ep)
        for k in range(3):
            sv = 1.0 if (w[6] >> k) & 1 else -1.0
            sw = 1.0 if (w[6] >> (3 + k)) & 1 else -1.0
            assert v[k] == float(np.float32(ar.init_velocity[k])) + (2.0 * (w[k] / 4294967296.0)) * sv
            assert om[k] == float(np.float32(ar.init_angular_velocity[k])) + (5.0 * (w[3 + k] / 4294967296.0)) * sw
    ar = qo.default_autoreset(seed=3)
    draws = np.array([np.concatenate(qo.reset_noise(ar, g, e)) for g in range(200) for e in range(10)])
    assert len({tuple(r) for r in draws}) == len(draws)
    assert np.all(np.abs(draws[:, :3]) < 2.0) and np.all(np.abs(draws[:, 3:]) < 5.0)
    assert abs(np.mean(draws > 0) - 0.5) < 0.02
    assert abs(np.mean(np.abs(draws[:, :3])) - 1.0) < 0.05 and abs(np.mean(np.abs(draws[:, 3:])) - 2.5) < 0.1
    other = np.concatenate(qo.reset_noise(qo.default_autoreset(seed=4), 0, 0))
    assert not np.array_equal(other, draws[0])


def test_autoreset_step_is_step_plus_reset():
    """qo_batch_env_step_autoreset == qo_batch_env_step, then for the finished envs zero state + the drawn noise
    (QuadrotorSim.reset quadrotorsim.py:239-258) and the observation of that state; ct stays as the done rule left it."""
    n, nt = 64, 4
    c = qo.default_consts(nt=nt)
    ar = qo.default_autoreset(seed=11, env_id_base=1000)
    rs = np.random.RandomState(0)
    mk = lambda: qo.make_states(np.zeros((n, 3), np.float32), rs2.uniform(-1, 1, (n, 3)), rs2.uniform(-2, 2, (n, 3)),
                                np.zeros((n, 4), np.float32), np.tile(np.eye(3, dtype=np.float32).reshape(9), (n, 1)))
    rs2 = np.random.RandomState(1)
    a_st = mk()
    rs2 = np.random.RandomState(1)
    b_st = mk()
    a_ct, b_ct, ep = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.uint32)
    for t in range(9):
        act = rs.uniform(0.1, 15, (n, 4)).astype(np.float32)
        obs_a, rew_a, done_a, fail_a = qo.batch_env_step_autoreset(c, ar, a_st, a_ct, ep, act)
        ep_before = ep - done_a.astype(np.uint32)
        obs_b, rew_b, done_b, fail_b = qo.batch_env_step(c, b_st, b_ct, act)
        assert np.array_equal(rew_a, rew_b) and np.array_equal(done_a, done_b) and np.array_equal(a_ct, b_ct)
        for e in np.nonzero(done_b)[0]:
            v, w = qo.reset_noise(ar, 1000 + e, ep_before[e])
            one = qo.make_states(np.zeros((1, 3), np.float32), v[None], w[None], np.zeros((1, 4), np.float32),
                                 np.eye(3, dtype=np.float32).reshape(1, 9))
            b_st[e] = one[0]
            obs_b[e] = qo.observe(c, one)[0]
        assert np.array_equal(obs_a, obs_b)
        sa, sb = qo.states_to_arrays(a_st), qo.states_to_arrays(b_st)
        for k in sa:
            assert np.array_equal(sa[k], sb[k]), k
    assert np.array_equal(ep, np.full(n, 2, np.uint32))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "quadrotor_traj_*.npz"))))
def test_numpy_pin_gap_is_bounded(path):
    """Every "bit-exact" statement here is against the reference under NumPy >= 2 (NEP 50); its own pin is numpy==1.22
    (requirements.txt:4), where a python float combined with a float32 SCALAR is float64 instead of float32
    (quadrotorsim.py:136-156 motor chain and thrust, env.py:217,237-241 reward). The oracle's legacy switch evaluates those
    mixes the 1.22 way (hand-derived, oracle/quadrotor_oracle.c). MEASURED gap between the two readings (tests/parity.py
    metric; profiles/r03/numpy_pin_gap.txt): within the north-star 1e-5 over the first 100 env steps (1 000 Euler sub-steps)
    of every golden rollout and over the whole near-hover / short rollouts (<= 4.1e-6); a craft tumbling under full-range
    random actions amplifies the half-ulp differences, and over 400-1 000 env steps the state gap reaches 1.5e-5 (8.3e-5 on
    the Euler angles of the observation) — the same size as the drift SURVEY.md App. C measured for an all-float32 rerun.
    So: parity with the NEP-50 run is parity with the pinned run to 1e-5 for 1 000 sub-steps, and to ~1e-4 at worst beyond."""
    g = np.load(path)
    new, _, _ = _rollout(g, path)
    qo.lib().qo_set_legacy_promotion(1)
    try:
        old, _, _ = _rollout(g, path)
    finally:
        qo.lib().qo_set_legacy_promotion(0)
    assert np.array_equal(new["done"], old["done"]) and np.array_equal(new["ct"], old["ct"])
    assert not np.array_equal(old["propw"], new["propw"])          # the switch does change the arithmetic

    def gap(k):
        e = {x: vec_rel_err(old[x][:k], new[x][:k]) for x in ("pos", "vel", "omega", "propw", "R")}
        e.update(obs=obs_rel_err(old["obs"][:k], new["obs"][:k]), reward=scalar_rel_err(old["reward"][:k], new["reward"][:k]),
                 power=scalar_rel_err(old["power"][:k], new["power"][:k]))
        return e
    first, whole = gap(100), gap(len(new["pos"]))
    print(os.path.basename(path), "numpy 1.22 vs NEP 50, first 100 env steps:", {k: "%.1e" % v for k, v in first.items()},
          "| all %d:" % len(new["pos"]), {k: "%.1e" % v for k, v in whole.items()})
    for k, v in first.items():
        assert v < REL_TOL, (k, v)
    for k, v in whole.items():
        assert v < (REL_TOL if ("hover" in path or "short" in path or "custom" in path) else 1e-4), (k, v)
