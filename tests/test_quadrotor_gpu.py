"""Parity of the HIP quadrotor kernel (through the C ABI / metagym_amd.Quadrotor) against
  (a) the CPU oracle (oracle/quadrotor_oracle.c) on identical inputs — bit-exact for everything that
      does not go through libm's atan2f, and
  (b) the golden vectors recorded from the unmodified reference — within the north-star 1e-5.
Runs on the GPU box only (-m gpu). Nothing here reads /root/reference."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import quadrotor as qo
from parity import REL_TOL, obs_rel_err, scalar_rel_err, vec_rel_err

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


CUSTOM_CONF = os.path.join(GOLDEN, "quadrotor_custom_config.json")


def _make_env(n, task="hovering_control", nt=1000, simulator_conf=None):
    import metagym_amd
    return metagym_amd.make("quadrotor-v0", num_envs=n, device="cuda:0", task=task, nt=nt,
                            simulator_conf=simulator_conf)


def _load_state(env, pos, vel, omega, propw, R, ct=None):
    sd = dict(pos=torch.as_tensor(np.ascontiguousarray(np.asarray(pos, np.float32).T)),
              vel=torch.as_tensor(np.ascontiguousarray(np.asarray(vel, np.float64).T)),
              omega=torch.as_tensor(np.ascontiguousarray(np.asarray(omega, np.float64).T)),
              propw=torch.as_tensor(np.ascontiguousarray(np.asarray(propw, np.float32).T)),
              rot=torch.as_tensor(np.ascontiguousarray(np.asarray(R, np.float32).reshape(len(pos), 9).T)),
              ct=torch.zeros(len(pos), dtype=torch.int32) if ct is None else torch.as_tensor(ct))
    env.load_state_dict(sd)


def _get_state(env):
    sd = env.state_dict()
    return dict(pos=sd["pos"].T.cpu().numpy(), vel=sd["vel"].T.cpu().numpy(), omega=sd["omega"].T.cpu().numpy(),
                propw=sd["propw"].T.cpu().numpy(), R=sd["rot"].T.cpu().numpy(), ct=sd["ct"].cpu().numpy())


def _assert_matches_oracle(gpu_state, gpu_out, st, ct, out):
    """gpu vs oracle on identical inputs: state, reward, done, failed bit-exact; obs bit-exact except
    the three atan2f angles (12..14), which get 4 ulp of slack."""
    o = qo.states_to_arrays(st)
    for k in ("pos", "vel", "omega", "propw", "R"):
        assert np.array_equal(gpu_state[k], o[k]), "state %s differs from the oracle" % k
    assert np.array_equal(gpu_state["ct"], ct)
    obs, rew, done, failed = out
    g_obs, g_rew64, g_done, g_failed = gpu_out
    assert np.array_equal(g_failed, failed.astype(np.uint8))
    assert np.array_equal(g_done, done.astype(bool))
    assert np.array_equal(g_rew64, rew)
    nonang = [i for i in range(16) if i not in (12, 13, 14)]
    assert np.array_equal(g_obs[:, nonang], obs[:, nonang])
    assert np.max(np.abs(g_obs[:, 12:15] - obs[:, 12:15])) <= 4 * np.spacing(np.float32(np.pi))


def test_single_step_matches_oracle_and_reference():
    g = np.load(os.path.join(GOLDEN, "quadrotor_onestep.npz"))
    n = len(g["actions"])
    env = _make_env(n)
    _load_state(env, g["in_pos"], g["in_vel"], g["in_omega"], g["in_propw"], g["in_R"])
    obs, rew, done, info = env.step(torch.as_tensor(g["actions"]))
    gs = _get_state(env)
    gpu_out = (obs.cpu().numpy(), env.reward64.cpu().numpy(), done.cpu().numpy(), info["failed"].cpu().numpy())
    # (a) oracle, bit-exact
    c = qo.default_consts()
    st = qo.make_states(g["in_pos"], g["in_vel"], g["in_omega"], g["in_propw"], g["in_R"])
    ct = np.zeros(n, np.int32)
    out = qo.batch_env_step(c, st, ct, g["actions"])
    _assert_matches_oracle(gs, gpu_out, st, ct, out)
    # (b) reference golden, north-star tolerance
    for k in ("pos", "vel", "omega", "propw", "R"):
        assert vec_rel_err(gs[k], g["out_" + k]) < REL_TOL, k
    assert obs_rel_err(gpu_out[0], g["obs"]) < REL_TOL
    assert scalar_rel_err(gpu_out[1], g["reward"]) < REL_TOL
    assert np.array_equal(gpu_out[2], g["done"])
    assert np.allclose(rew.cpu().numpy(), g["reward"].astype(np.float32), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "quadrotor_traj_*.npz"))))
def test_rollout_matches_reference_golden(path):
    """The whole recorded episode (up to 1000 env steps = 10 000 Euler sub-steps) stays within 1e-5
    of the unmodified reference; ct/done are exact. Uses step() for the first half and the
    multi-step rollout() launch for the second half."""
    g = np.load(path)
    T = len(g["reward"])
    # *_custom_*: recorded with the non-stock simulator config -> the general kernel specialisation
    env = _make_env(1, nt=int(g["nt"]), simulator_conf=CUSTOM_CONF if "_custom_" in os.path.basename(path) else None)
    obs0 = env.reset(init_velocity=g["init_vel"][None], init_angular_velocity=g["init_omega"][None])
    assert obs_rel_err(obs0.cpu().numpy(), g["obs0"][None]) < REL_TOL
    acts = torch.as_tensor(g["actions"][:T]).cuda()
    half = T // 2
    rec = {k: [] for k in ("pos", "vel", "omega", "propw", "R", "obs", "reward", "done", "ct")}
    for t in range(half):
        obs, rew, done, info = env.step(acts[t][None])
        s = _get_state(env)
        for k in ("pos", "vel", "omega", "propw", "R"):
            rec[k].append(s[k][0])
        rec["ct"].append(int(s["ct"][0]))
        rec["obs"].append(obs[0].cpu().numpy())
        rec["reward"].append(float(env.reward64[0]))
        rec["done"].append(bool(done[0]))
        assert int(info["failed"][0]) == 0
    obs, rew, done, failed = env.rollout(acts[half:, None, :])
    assert int(failed.max()) == 0
    r_obs = np.concatenate([np.asarray(rec["obs"]), obs[:, 0].cpu().numpy()])
    r_rew = np.concatenate([np.asarray(rec["reward"]), env._last_rollout_reward64[:, 0].cpu().numpy()])
    r_done = np.concatenate([np.asarray(rec["done"]), done[:, 0].cpu().numpy()])
    assert np.array_equal(r_done, g["done"])
    assert np.array_equal(np.asarray(rec["ct"]), g["ct"][:half])
    final = _get_state(env)
    assert int(final["ct"][0]) == int(g["ct"][-1])
    errs = dict(obs=obs_rel_err(r_obs, g["obs"]), reward=scalar_rel_err(r_rew, g["reward"]))
    for k in ("pos", "vel", "omega", "propw", "R"):
        errs[k] = max(vec_rel_err(np.asarray(rec[k]), g[k][:half]), vec_rel_err(final[k][0][None], g[k][-1][None]))
    print(os.path.basename(path), {k: "%.2e" % v for k, v in errs.items()})
    for k, v in errs.items():
        assert v < REL_TOL, (k, v)


def test_failure_flags_match_reference():
    g = np.load(os.path.join(GOLDEN, "quadrotor_fail.npz"))
    n = len(g["failed"])
    env = _make_env(n)
    _load_state(env, g["in_pos"], g["in_vel"], g["in_omega"], g["in_propw"], g["in_R"])
    obs, rew, done, info = env.step(torch.as_tensor(g["actions"]))
    failed = info["failed"].cpu().numpy()
    assert np.array_equal(failed != 0, g["failed"])
    assert list(failed[:4]) == [1, 1, 2, 3]          # range, range, velocity, body rate
    assert done.cpu().numpy()[g["failed"]].all()
    assert (rew.cpu().numpy()[g["failed"]] == 0).all()
    s = _get_state(env)
    for k in ("pos", "vel", "omega", "propw", "R"):   # frozen at the failing sub-step, like the raise
        assert vec_rel_err(s[k], g["out_" + k]) < REL_TOL, k


def _random_batch(n, seed):
    rs = np.random.RandomState(seed)
    pos = (rs.uniform(-30, 30, (n, 3)) * [1, 1, 0.15]).astype(np.float32)
    vel = rs.uniform(-4, 4, (n, 3))
    omega = rs.uniform(-5, 5, (n, 3))
    propw = rs.uniform(0, 600, (n, 4)).astype(np.float32)
    R = np.tile(np.eye(3, dtype=np.float32).reshape(9), (n, 1))
    R += rs.uniform(-0.05, 0.05, (n, 9)).astype(np.float32)
    return pos, vel, omega, propw, R


def test_batch_matches_oracle_bitexact_multi_step():
    """4096 random envs x 12 steps, ragged size (not a multiple of the wave/block size)."""
    n, T = 4096 + 37, 12
    pos, vel, omega, propw, R = _random_batch(n, 7)
    env = _make_env(n, nt=5)                      # nt=5 exercises the ct wrap inside the window
    _load_state(env, pos, vel, omega, propw, R)
    c = qo.default_consts(nt=5)
    st = qo.make_states(pos, vel, omega, propw, R)
    ct = np.zeros(n, np.int32)
    acts = np.random.RandomState(8).uniform(-0.5, 15.5, (T, n, 4)).astype(np.float32)
    for t in range(T):
        obs, rew, done, info = env.step(torch.as_tensor(acts[t]))
        out = qo.batch_env_step(c, st, ct, acts[t])
        gpu_out = (obs.cpu().numpy(), env.reward64.cpu().numpy(), done.cpu().numpy(), info["failed"].cpu().numpy())
        _assert_matches_oracle(_get_state(env), gpu_out, st, ct, out)
    assert done.any() or True


def test_custom_config_batch_matches_oracle_bitexact():
    """Non-stock simulator config (off-diagonal inertia, shifted centre of gravity, asymmetric
    propellers, 5 sub-steps): the general kernel specialisation against the oracle, 2048 envs x 6 steps."""
    import json
    n, T = 2048, 6
    with open(CUSTOM_CONF) as f:
        c = qo.consts_from_config(json.load(f))
    pos, vel, omega, propw, R = _random_batch(n, 21)
    env = _make_env(n, simulator_conf=CUSTOM_CONF)
    _load_state(env, pos, vel, omega, propw, R)
    st = qo.make_states(pos, vel, omega, propw, R)
    ct = np.zeros(n, np.int32)
    rs = np.random.RandomState(5)
    for t in range(T):
        a = rs.uniform(0.0, 15.0, (n, 4)).astype(np.float32)
        obs, rew, done, info = env.step(torch.as_tensor(a))
        out = qo.batch_env_step(c, st, ct, a)
        gpu_out = (obs.cpu().numpy(), env.reward64.cpu().numpy(), done.cpu().numpy(), info["failed"].cpu().numpy())
        _assert_matches_oracle(_get_state(env), gpu_out, st, ct, out)


def test_rollout_equals_repeated_step():
    n, T = 1000, 9
    pos, vel, omega, propw, R = _random_batch(n, 11)
    acts = torch.as_tensor(np.random.RandomState(12).uniform(0.1, 15, (T, n, 4)).astype(np.float32)).cuda()
    a = _make_env(n, nt=4)
    b = _make_env(n, nt=4)
    _load_state(a, pos, vel, omega, propw, R)
    _load_state(b, pos, vel, omega, propw, R)
    obs_r, rew_r, done_r, failed_r = a.rollout(acts)
    for t in range(T):
        obs, rew, done, info = b.step(acts[t])
        assert torch.equal(obs, obs_r[t]) and torch.equal(rew, rew_r[t]) and torch.equal(done, done_r[t])
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        if torch.is_tensor(sa[k]):
            assert torch.equal(sa[k], sb[k]), k


def test_collision_with_obstacle_map(tmp_path):
    """no_collision task with a real map file: np.any() over the python-sliced AABB (env.py:248-260)."""
    grid = np.zeros((20, 20), dtype=int)
    grid[5, 5] = -1
    grid[5, 6] = 3       # an obstacle right next to the start cell
    grid[0, :] = 1
    p = tmp_path / "map.txt"
    p.write_text("\n".join(" ".join(str(v) for v in row) for row in grid))
    import metagym_amd
    n = 64
    env = metagym_amd.make("quadrotor-v0", num_envs=n, device="cuda:0", task="no_collision", map_file=str(p))
    rs = np.random.RandomState(3)
    pos = np.zeros((n, 3), np.float32)
    pos[:, 0] = rs.uniform(-7, 3, n)     # some start left of the map edge (negative index wrap)
    pos[:, 1] = rs.uniform(-7, 3, n)
    pos[:, 2] = rs.uniform(-5.5, -3.0, n)   # around z+5 in [-0.5, 2]
    vel = rs.uniform(-3, 3, (n, 3))
    omega = np.zeros((n, 3))
    propw = np.zeros((n, 4), np.float32)
    R = np.tile(np.eye(3, dtype=np.float32).reshape(9), (n, 1))
    _load_state(env, pos, vel, omega, propw, R)
    acts = rs.uniform(0.1, 15, (n, 4)).astype(np.float32)
    obs, rew, done, info = env.step(torch.as_tensor(acts))
    c = qo.default_consts(task=qo.TASK_NO_COLLISION)
    m = env.map_matrix.astype(np.int32)
    import ctypes as C
    c.map = m.ctypes.data_as(C.POINTER(C.c_int32))
    c.map_h, c.map_w = m.shape
    c.x_offset, c.y_offset = env.x_offset, env.y_offset
    st = qo.make_states(pos, vel, omega, propw, R)
    ct = np.zeros(n, np.int32)
    o_obs, o_rew, o_done, o_failed = qo.batch_env_step(c, st, ct, acts)
    assert np.array_equal(done.cpu().numpy(), o_done.astype(bool))
    assert np.array_equal(env.reward64.cpu().numpy(), o_rew)
    assert 0 < o_done.sum() < n      # the case set really contains both outcomes


def test_full_size_properties_65536():
    """BASELINE config C2 size (65 536 envs): size-independent properties instead of an oracle sweep —
    determinism (same inputs twice -> identical bits), permutation equivariance (env order is
    irrelevant: no cross-env coupling), and a checkpoint round trip."""
    n, T = 65536, 5
    a = _make_env(n)
    b = _make_env(n)
    a.reset(seed=123)
    b.reset(seed=123)
    acts = torch.rand(T, n, 4, device="cuda:0") * 14.9 + 0.1
    perm = torch.randperm(n, device="cuda:0")
    c = _make_env(n)
    sd = a.state_dict()
    c.load_state_dict({k: (v[..., perm] if v.dim() > 1 else v[perm]) for k, v in sd.items() if torch.is_tensor(v)})
    for t in range(T):
        oa, ra, da, _ = a.step(acts[t])
        ob, rb, db, _ = b.step(acts[t])
        oc, rc, dc, _ = c.step(acts[t][perm])
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db)
        assert torch.equal(oa[perm], oc) and torch.equal(ra[perm], rc) and torch.equal(da[perm], dc)
    assert torch.isfinite(oa).all()
    # spot-check 256 random envs of the big batch against the oracle, bit-exact state
    idx = np.random.RandomState(0).choice(n, 256, replace=False)
    s0 = {k: v.cpu().numpy() for k, v in sd.items() if torch.is_tensor(v)}
    st = qo.make_states(s0["pos"].T[idx], s0["vel"].T[idx], s0["omega"].T[idx], s0["propw"].T[idx], s0["rot"].T[idx])
    ct = np.zeros(256, np.int32)
    cc = qo.default_consts()
    acts_h = acts.cpu().numpy()
    for t in range(T):
        qo.batch_env_step(cc, st, ct, acts_h[t][idx])
    o = qo.states_to_arrays(st)
    fin = _get_state(a)
    for k in ("pos", "vel", "omega", "propw", "R"):
        assert np.array_equal(fin[k][idx], o[k]), k


def test_reset_reproduces_reference_rng_order():
    """env 0 of reset(seed=s) equals `np.random.seed(s); env.reset()` of the reference (golden init_*)."""
    for name, seed in (("full_s0", 0), ("full_s1", 1), ("hover_s2", 2)):
        g = np.load(os.path.join(GOLDEN, "quadrotor_traj_%s.npz" % name))
        env = _make_env(3)
        obs = env.reset(seed=seed)
        s = _get_state(env)
        assert np.array_equal(s["vel"][0], g["init_vel"])
        assert np.array_equal(s["omega"][0], g["init_omega"])
        assert obs_rel_err(obs[:1].cpu().numpy(), g["obs0"][None]) < REL_TOL


def test_masked_reset_only_touches_selected_envs():
    n = 300
    env = _make_env(n)
    env.reset(seed=5)
    acts = torch.full((n, 4), 3.0, device="cuda:0")
    for _ in range(3):
        env.step(acts)
    before = env.state_dict()
    mask = torch.zeros(n, dtype=torch.bool)
    mask[::7] = True
    env.reset(mask=mask, seed=6)
    after = env.state_dict()
    keep = ~mask.cuda()
    for k in ("pos", "vel", "omega", "propw", "rot"):
        assert torch.equal(before[k][:, keep], after[k][:, keep]), k
    assert (after["pos"][:, mask.cuda()] == 0).all()
    assert torch.equal(before["ct"], after["ct"])     # reset() does not clear ct (env.py:116-125)


def test_no_collision_on_reference_map_matches_golden(tmp_path):
    """The reference's own obstacle map (recorded in the golden file) through map_file=...: done and
    position sequences exact, reward within 1e-5 of the reference."""
    import metagym_amd
    g = np.load(os.path.join(GOLDEN, "quadrotor_no_collision_map.npz"))
    p = tmp_path / "map.txt"
    p.write_text("\n".join(" ".join("%d" % v for v in row) for row in g["map"]))
    n = 3
    env = metagym_amd.make("quadrotor-v0", num_envs=n, device="cuda:0", task="no_collision", map_file=str(p),
                           nt=int(g["nt"]))
    iv = np.stack([g["init_vel_%d" % k] for k in range(n)])
    iw = np.stack([g["init_omega_%d" % k] for k in range(n)])
    env.reset(init_velocity=iv, init_angular_velocity=iw)
    lens = [len(g["actions_%d" % k]) for k in range(n)]
    seen_done = False
    for t in range(max(lens)):
        a = np.stack([g["actions_%d" % k][t] if t < lens[k] else np.full(4, 2.0, np.float32) for k in range(n)])
        obs, rew, done, info = env.step(torch.as_tensor(a))
        pos = env.pos.T.cpu().numpy()
        for k in range(n):
            if t >= lens[k]:
                continue            # this recorded trajectory is shorter; the env just idles
            assert bool(done[k]) == bool(g["done_%d" % k][t]), (k, t)
            assert np.array_equal(pos[k], g["pos_%d" % k][t]), (k, t)
            assert abs(float(env.reward64[k]) - g["reward_%d" % k][t]) <= 1e-5 * max(1.0, abs(g["reward_%d" % k][t]))
            seen_done = seen_done or bool(done[k])
    assert seen_done


def test_velocity_control_matches_oracle_and_reference():
    """velocity_control: the GPU-rolled target trajectory is bit-identical to the oracle's (pure f32,
    no libm) and within 1e-5 of the reference's; a 260-step rollout over ct == nt matches the
    reference golden (19-entry obs, reward, done) and auto-reset keeps stepping."""
    import ctypes as C
    import metagym_amd
    g = np.load(os.path.join(GOLDEN, "quadrotor_velocity_control.npz"))
    nt, seed = int(g["nt"]), int(g["seed"])
    n = 5
    env = metagym_amd.make("quadrotor-v0", num_envs=n, device="cuda:0", task="velocity_control", nt=nt, seed=seed)
    tg = env.velocity_targets.cpu().numpy()
    c = qo.default_consts(nt=nt, task=qo.TASK_VELOCITY)
    c.x_offset = c.y_offset = 0
    c.z_offset = 0.0
    otg = qo.velocity_targets(c, qo.velocity_target_actions(seed, nt))
    assert np.array_equal(tg, otg)
    assert np.array_equal(tg, g["targets"])        # bit-identical to the reference's trajectory
    iv = np.tile(g["init_vel"], (n, 1))
    iw = np.tile(g["init_omega"], (n, 1))
    obs0 = env.reset(init_velocity=iv, init_angular_velocity=iw)
    assert obs0.shape == (n, 19)
    assert obs_rel_err(obs0[:1, :16].cpu().numpy(), g["obs0"][None, :16], z_offset=1.0) < REL_TOL
    assert np.array_equal(obs0[0, 16:].cpu().numpy(), tg[0])
    c.velocity_targets = otg.ctypes.data_as(C.POINTER(C.c_float))
    st = qo.make_states(np.zeros((1, 3), np.float32), g["init_vel"][None], g["init_omega"][None],
                        np.zeros((1, 4), np.float32), np.eye(3, dtype=np.float32).reshape(1, 9))
    ct = C.c_int(0)
    for t in range(len(g["actions"])):
        a = np.tile(g["actions"][t], (n, 1))
        obs, rew, done, info = env.step(torch.as_tensor(a))
        o = obs.cpu().numpy()
        assert np.array_equal(o[0], o[n - 1])                      # identical envs stay identical
        assert bool(done[0]) == bool(g["done"][t]), t
        assert obs_rel_err(o[:1, :16], g["obs"][t][None, :16], z_offset=1.0) < REL_TOL, t
        assert vec_rel_err(o[:1, 16:], g["obs"][t][None, 16:]) < REL_TOL, t
        assert scalar_rel_err(float(env.reward64[0]), g["reward"][t]) < REL_TOL, t
        oo, r, d, f = qo.env_step_velocity(c, st[0], ct, g["actions"][t])
        assert float(env.reward64[0]) == r and np.array_equal(o[0, 16:], oo[16:]), t    # vs oracle: exact
        assert "next_target_g_v_x" in info and len(info) == 20
    env2 = metagym_amd.make("quadrotor-v0", num_envs=64, device="cuda:0", task="velocity_control", nt=20, seed=1,
                            auto_reset=True)
    env2.reset(seed=0)
    dones = 0
    for t in range(45):
        _, _, d, _ = env2.step(torch.full((64, 4), 2.2))
        dones += int(d.sum())
    assert dones == 2 * 64                                         # ct == nt twice per env


# ---- fused auto-reset: the launch bench.py times (mg_quadrotor_plan_step with an auto-reset block) ---------

RANDOM123_KAT = [   # Random123 kat_vectors, philox4x32 10: counter[4], key[2] -> output[4]
    ([0x00000000] * 4, [0x00000000] * 2, [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


def test_device_philox_known_answers():
    """The device generator behind every fused auto-reset against the published Random123 vectors, and against
    the oracle's restatement on 4096 random (counter, key) pairs."""
    from metagym_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(0)
    rnd = rs.randint(0, 2 ** 32, (4096, 6), dtype=np.uint64).astype(np.uint32)
    inp = np.concatenate([np.array([c + k for c, k, _ in RANDOM123_KAT], np.uint32), rnd])
    d_in = torch.as_tensor(inp.view(np.int32)).cuda()
    d_out = torch.zeros(len(inp), 4, dtype=torch.int32, device="cuda:0")
    rc = lib.mg_selftest_philox(_lib.ptr(d_in), _lib.ptr(d_out), len(inp), _lib.current_stream(d_out.device))
    _lib.check(rc, "mg_selftest_philox")
    out = d_out.cpu().numpy().view(np.uint32)
    for i, (_, _, want) in enumerate(RANDOM123_KAT):
        assert [int(x) for x in out[i]] == want, i
    for i in range(3, len(inp), 97):
        assert [int(x) for x in out[i]] == qo.philox4x32_10(inp[i, :4], inp[i, 4:]), i


def _autoreset_env(n, nt, seed, env_id_base=0, task="hovering_control"):
    import metagym_amd
    return metagym_amd.make("quadrotor-v0", num_envs=n, device="cuda:0", task=task, nt=nt, auto_reset=True,
                            seed=seed, env_id_base=env_id_base)


def _hard_actions(rs, T, n):
    """Mostly full-range voltages, with stretches of idle motors so that some envs drop through the floor
    (collision episode ends) besides the ct == nt ones."""
    a = rs.uniform(0.1, 15.0, (T, n, 4)).astype(np.float32)
    idle = rs.random_sample((T, n)) < 0.25
    a[idle] = 0.1
    return a


def test_autoreset_matches_oracle_bit_exact():
    """(i) The benched launch against the oracle's restatement of it: 4 133 envs, >= 2 episode ends per env
    (ct == nt twice, plus floor collisions), every step's obs / reward / done / failed, the final state and the
    per-env episode counters; 64-bit seed and a non-zero env_id_base."""
    n, nt, T = 4096 + 37, 23, 52
    seed, base = (0xC0FFEE << 32) | 0x1234567, (1 << 33) + 12345
    env = _autoreset_env(n, nt, seed, env_id_base=base)
    env.reset(seed=3)
    env.pos[2, ::3] = -4.97          # a third of the envs start 3 cm above the floor: collision episode ends
    sd = {k: v.cpu().numpy() for k, v in env.state_dict().items() if torch.is_tensor(v)}
    st = qo.make_states(sd["pos"].T, sd["vel"].T, sd["omega"].T, sd["propw"].T, sd["rot"].T)
    ct = np.zeros(n, np.int32)
    ep = np.zeros(n, np.uint32)
    c = qo.default_consts(nt=nt)
    ar = qo.default_autoreset(seed=seed, env_id_base=base)
    acts = _hard_actions(np.random.RandomState(4), T, n)
    ends = np.zeros(n, int)
    collided = 0
    for t in range(T):
        obs, rew, done, info = env.step(torch.as_tensor(acts[t]))
        ct_before = ct.copy()
        out = qo.batch_env_step_autoreset(c, ar, st, ct, ep, acts[t])
        gpu_out = (obs.cpu().numpy(), env.reward64.cpu().numpy(), done.cpu().numpy(), info["failed"].cpu().numpy())
        _assert_matches_oracle(_get_state(env), gpu_out, st, ct, out)
        ends += out[2]
        collided += int(np.sum((out[2] != 0) & (ct_before + 1 != nt)))
    assert ends.min() >= 2 and collided > 0
    assert np.array_equal(env.episode.cpu().numpy().view(np.uint32), ep) and np.array_equal(ep, ends.astype(np.uint32))


def test_autoreset_equals_explicit_masked_reset_with_the_drawn_noise():
    """(ii) auto_reset=True  ==  a twin env stepped without it whose finished envs are reset by hand with
    reset(mask=done, init_velocity=<the Philox draws of that env's k-th restart>): same observations (the reset
    row is the first observation of the next episode), same state, same counters."""
    import metagym_amd
    n, nt, T, seed = 777, 9, 31, 99
    auto = _autoreset_env(n, nt, seed)
    twin = metagym_amd.make("quadrotor-v0", num_envs=n, device="cuda:0", task="hovering_control", nt=nt, seed=seed)
    auto.reset(seed=1)
    twin.reset(seed=1)
    ar = qo.default_autoreset(seed=seed)
    acts = _hard_actions(np.random.RandomState(2), T, n)
    episodes = np.zeros(n, int)
    for t in range(T):
        a = torch.as_tensor(acts[t]).cuda()
        oa, ra, da, ia = auto.step(a)
        ot, rt, dt_, it = twin.step(a)
        ot = ot.clone()
        assert torch.equal(ra, rt) and torch.equal(da, dt_) and torch.equal(ia["failed"], it["failed"])
        d = dt_.cpu().numpy()
        if d.any():
            iv, iw = np.zeros((n, 3)), np.zeros((n, 3))
            for e in np.nonzero(d)[0]:
                iv[e], iw[e] = qo.reset_noise(ar, e, episodes[e])
                episodes[e] += 1
            ot_reset = twin.reset(mask=dt_, init_velocity=iv, init_angular_velocity=iw)
            ot[dt_] = ot_reset[dt_]
        assert torch.equal(oa, ot), t
        sa, stw = auto.state_dict(), twin.state_dict()
        for k in ("pos", "vel", "omega", "propw", "rot", "ct"):
            assert torch.equal(sa[k], stw[k]), (k, t)
    assert episodes.min() >= 2
    assert np.array_equal(auto.episode.cpu().numpy(), episodes)
    assert int(twin.episode.abs().sum()) == 0            # the explicit path never touches the counters


def test_autoreset_is_shard_invariant():
    """(iii) Two half-shards with env_id_base 0 / n/2 == one full batch: the noise of an env depends on its global
    id and its own episode count only (multi-GPU sharding cannot change a trajectory)."""
    n, nt, T, seed = 1024, 7, 30, 5
    full = _autoreset_env(n, nt, seed)
    lo = _autoreset_env(n // 2, nt, seed, env_id_base=0)
    hi = _autoreset_env(n // 2, nt, seed, env_id_base=n // 2)
    full.reset(seed=8)
    sd = full.state_dict()
    for env, sl in ((lo, slice(0, n // 2)), (hi, slice(n // 2, n))):
        env.load_state_dict({k: (v[..., sl] if v.dim() > 1 else v[sl]) for k, v in sd.items() if torch.is_tensor(v)})
    acts = torch.as_tensor(_hard_actions(np.random.RandomState(6), T, n)).cuda()
    for t in range(T):
        of, rf, df, _ = full.step(acts[t])
        ol, rl, dl, _ = lo.step(acts[t, : n // 2].contiguous())
        oh, rh, dh, _ = hi.step(acts[t, n // 2:].contiguous())
        assert torch.equal(of, torch.cat([ol, oh])) and torch.equal(rf, torch.cat([rl, rh]))
        assert torch.equal(df, torch.cat([dl, dh]))
    assert int(full.episode.min()) >= 2
    sf, sl_, sh = full.state_dict(), lo.state_dict(), hi.state_dict()
    for k in ("pos", "vel", "omega", "propw", "rot", "ct", "episode"):
        assert torch.equal(sf[k], torch.cat([sl_[k], sh[k]], dim=-1)), k


def test_autoreset_rollout_equals_single_steps():
    """(iv) n_steps > 1 with resets INSIDE the launch == the same steps one launch at a time (the noise no longer
    depends on a step counter, so the two launch shapes must agree bit for bit)."""
    n, nt, T, seed = 1500, 6, 20, 77
    a, b = _autoreset_env(n, nt, seed), _autoreset_env(n, nt, seed)
    a.reset(seed=4)
    b.reset(seed=4)
    acts = torch.as_tensor(_hard_actions(np.random.RandomState(9), T, n)).cuda()
    obs_r, rew_r, done_r, failed_r = a.rollout(acts)
    for t in range(T):
        obs, rew, done, info = b.step(acts[t])
        assert torch.equal(obs, obs_r[t]) and torch.equal(rew, rew_r[t]) and torch.equal(done, done_r[t]), t
        assert torch.equal(info["failed"], failed_r[t])
    assert int(done_r.sum(0).min()) >= 3
    sa, sb = a.state_dict(), b.state_dict()
    for k in ("pos", "vel", "omega", "propw", "rot", "ct", "episode"):
        assert torch.equal(sa[k], sb[k]), k


def test_autoreset_full_size_sampled_against_oracle():
    """The C2 batch itself (65 536 envs, the size bench.py times): 256 sampled envs of the fused-auto-reset run
    against the oracle, bit-exact state and counters, plus the noise range / sign statistics of every restart."""
    n, nt, T, seed = 65536, 11, 25, 1000
    env = _autoreset_env(n, nt, seed)
    env.reset(seed=1000)
    sd = {k: v.cpu().numpy() for k, v in env.state_dict().items() if torch.is_tensor(v)}
    idx = np.sort(np.random.RandomState(0).choice(n, 256, replace=False))
    acts = torch.rand(T, n, 4, device="cuda:0") * 14.9 + 0.1
    acts_h = acts.cpu().numpy()
    first_obs = []
    for t in range(T):
        obs, rew, done, info = env.step(acts[t])
        if t == nt - 1:
            assert bool(done.all())                                  # every env just restarted (ct == nt) ...
            first_obs.append(obs.cpu().numpy().copy())               # ... so this is 65 536 reset observations
    c = qo.default_consts(nt=nt)
    fin = _get_state(env)
    ep = env.episode.cpu().numpy().view(np.uint32)
    for j, e in enumerate(idx):
        st = qo.make_states(sd["pos"].T[[e]], sd["vel"].T[[e]], sd["omega"].T[[e]], sd["propw"].T[[e]], sd["rot"].T[[e]])
        ct, epo = np.zeros(1, np.int32), np.zeros(1, np.uint32)
        ar = qo.default_autoreset(seed=seed, env_id_base=int(e))
        for t in range(T):
            qo.batch_env_step_autoreset(c, ar, st, ct, epo, acts_h[t][[e]])
        o = qo.states_to_arrays(st)
        for k in ("pos", "vel", "omega", "propw", "R"):
            assert np.array_equal(fin[k][e], o[k][0]), (k, e)
        assert fin["ct"][e] == ct[0] and ep[e] == epo[0]
    # reset observations: b_v = velocity (R = I), gyro = body rate; |v| < 2, |w| < 5 per component, signs balanced
    o = first_obs[0]
    assert np.all(np.abs(o[:, 0:3]) <= 2.0) and np.all(np.abs(o[:, 9:12]) <= 5.0)
    assert np.all(o[:, 3:6] == 0) and np.all(o[:, 15] == 5.0)
    for cols, scale in ((slice(0, 3), 2.0), (slice(9, 12), 5.0)):
        x = o[:, cols] / scale
        assert abs(np.mean(x > 0) - 0.5) < 0.01 and abs(np.mean(np.abs(x)) - 0.5) < 0.01


def test_step_outputs_alias_unless_copy_outputs():
    """step() returns the same four tensors every call (no allocation on the hot path, hipGraph-replayable);
    `copy_outputs=True` gives fresh ones like the reference's fresh numpy arrays."""
    import metagym_amd
    a = torch.full((8, 4), 2.0, device="cuda:0")
    for copy in (False, True):
        env = metagym_amd.make("quadrotor-v0", num_envs=8, device="cuda:0", task="hovering_control", copy_outputs=copy)
        env.reset(seed=0)
        o1, r1, d1, i1 = env.step(a)
        keep = o1.clone()
        o2, r2, d2, i2 = env.step(a)
        assert (o1.data_ptr() == o2.data_ptr()) == (not copy)
        assert torch.equal(o1, keep) == copy                      # without copies the first observation was overwritten
        assert torch.equal(i2["z"], o2[:, 15])
