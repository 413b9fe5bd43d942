"""HIP walker engine (through the C ABI / metagym_amd.metalocomotion) against the numpy oracle
(oracle/abd.py) on identical inputs, plus invariants at the BASELINE C4 batch size.
Physics parity with the reference is UNPINNED (PyBullet is not in the reference tree): what is
checked is GPU == oracle to float64 round-off, the pinned Python-side rules (obs layout, reward
terms, done rule), and physical invariants. GPU box only (-m gpu)."""
import copy
import os

import numpy as np
import pytest
import torch

from oracle import abd
from walker_fixtures import load_models, preset_of, world_kw

pytestmark = pytest.mark.gpu
MODELS = load_models()
RULES = os.path.join(os.path.dirname(__file__), "golden", "walker_rules.npz")


def _make(cls_name, models, n, task_ids=None, **kw):
    import metagym_amd.metalocomotion as ml
    kw.setdefault("preset", preset_of(models[0]))         # the world (body damping, velocity clamp) the models were read for
    env = getattr(ml, cls_name)(num_envs=n, device="cuda:0", **kw)
    env.set_task(models, task_ids)
    return env


def _oracle_env(m, ant=False, **kw):
    if ant:
        return abd.WalkerEnv(m, prm=abd.Params(friction=0.8 * float(m.geom_friction), power=2.5,
                                               self_friction=float(m.geom_friction) ** 2, **world_kw(m)),
                             motor_power=np.full(len(m.joint_lo), 100.0), alive_z=0.26, alive_bonus=1.0,
                             initial_z=None, torque_f32=False, **kw)
    return abd.WalkerEnv(m, prm=abd.Params(friction=0.8 * float(m.geom_friction),
                                           self_friction=float(m.geom_friction) ** 2, **world_kw(m)), **kw)


@pytest.mark.parametrize("mapping", ["wave", "lane"])
@pytest.mark.parametrize("robot", ["humanoid", "ant"])
def test_gpu_matches_oracle_trajectory(robot, mapping):
    """4 body variants x 3 envs, 25 env steps (100 physics sub-steps incl. landing on the ground and
    joint-limit pushes): state, obs, reward terms and done agree with the oracle."""
    ant = robot == "ant"
    names = ["ant", "ant_tra_005"] if ant else ["humanoid", "humanoid_tra_000", "humanoid_tra_137", "humanoid_ood_003"]
    models = [MODELS[k] for k in names]
    n = 3 * len(models)
    env = _make("MetaAntEnv" if ant else "MetaHumanoidEnv", models, n, max_steps=20, mapping=mapping)
    ids = env.task_id.cpu().numpy()
    nj = env.n_joints
    rs = np.random.RandomState(0)
    noise = rs.uniform(-0.1, 0.1, (n, nj))
    obs = env.reset(joint_noise=noise).cpu().numpy()
    oenvs = [_oracle_env(models[ids[e]], ant, max_steps=20) for e in range(n)]
    for e in range(n):
        o = oenvs[e].reset(noise[e])
        assert np.allclose(obs[e], o, rtol=0, atol=1e-6), e
    worst = 0.0
    for t in range(25):
        a = rs.uniform(-1.3, 1.3, (n, nj)).astype(np.float32)
        obs, rew, done, info = env.step(torch.as_tensor(a))
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        r5 = info["rewards"].cpu().numpy()
        q = env.q.cpu().numpy().T
        pos = env.pos.cpu().numpy().T
        for e in range(n):
            o, r, d, inf = oenvs[e].step(a[e])
            s = oenvs[e].s
            worst = max(worst, np.abs(q[e] - s.q).max(), np.abs(pos[e] - s.pos).max())
            assert np.allclose(q[e], s.q, rtol=0, atol=1e-7), (t, e)
            assert np.allclose(pos[e], s.pos, rtol=0, atol=1e-7), (t, e)
            assert np.allclose(obs[e], o, rtol=0, atol=2e-5), (t, e, np.abs(obs[e] - o).max())
            assert np.allclose(r5[e], inf["rewards"], rtol=1e-5, atol=1e-4), (t, e)
            assert abs(rew[e] - r) < 1e-4 * max(1.0, abs(r)), (t, e)
            assert bool(done[e]) == d, (t, e)
            assert int(info["steps"][e]) == inf["steps"]
    print(robot, mapping, "max |state diff| GPU vs oracle over 25 steps: %.2e" % worst)


@pytest.mark.parametrize("mapping", ["wave", "lane"])
@pytest.mark.parametrize("c", range(6))
def test_gpu_reproduces_what_the_reference_computed(c, mapping):
    """The HIP kernels against tests/golden/walker_rules.npz — observations, reward terms, done, steps and feet
    flags the UNMODIFIED reference Python computed (on oracle/abd.py's dynamics, see oracle/refstubs/pybullet),
    for 4 humanoid + 2 ant variants, >= 50 steps each, incl. a fall, a max_steps cut and second resets. The env is
    built from the reference's task FILE NAME (regenerated variant), reset with the recorded joint noise and driven
    with the recorded float32 actions. float32 observations within 1e-6 (values reach 5: that is 2 ulp), reward
    terms within 1e-5 relative (the kernels hand them back as float32), flags exact; simulator state within 1e-6
    of the recorded float64 state (GPU vs numpy round-off of the same engine — L4 itself stays unpinned).
    The first 50 steps of every episode run freely. Beyond that (only the 140-step base-humanoid run, which has
    fallen over and tumbles on the ground by then) the round-off between the two implementations of the engine is
    amplified by the contact dynamics past the float32 resolution of the observation, which says nothing about the
    rules under test: from step 50 on the recorded state is loaded before every step, so each step is checked on its
    own."""
    import metagym_amd.metalocomotion as ml
    g = np.load(RULES)
    k = "case%d_" % c
    case = {n[len(k):]: g[n] for n in g.files if n.startswith(k)}
    task = str(case["task"])
    cls = ml.MetaHumanoidEnv if task.startswith("humanoid") else ml.MetaAntEnv
    n = 3                                             # three identical envs: lanes / waves must agree with each other too
    # preset="mujoco": the golden was recorded (round 2) with the stand-in dynamics on MuJoCo's reading of the files; what it
    # pins — the reference's Python-side rules — is the same under either reading
    env = cls(num_envs=n, device="cuda:0", max_steps=int(case["max_steps"]), mapping=mapping, preset="mujoco")
    assert task in env.tra_tasks + env.tst_tasks + env.ood_tasks or task in ("humanoid.xml", "ant.xml")
    env.set_task(task)
    assert env.models[0].joint_names == [str(x) for x in case["joint_names"]]
    t = 0
    worst = dict(obs=0.0, state=0.0)
    for ep, T in enumerate(case["episode_lengths"]):
        obs0 = env.reset(joint_noise=np.tile(case["reset_joint_noise"][ep], (n, 1))).cpu().numpy()
        assert np.max(np.abs(obs0 - case["reset_obs"][ep])) <= 1e-6, ("reset obs", ep)
        pot = env.potential.cpu().numpy()
        assert np.max(np.abs(pot - case["reset_potential"][ep])) <= 1e-9 * abs(case["reset_potential"][ep])
        for i in range(int(T)):
            if i >= 50:                               # re-synchronise: state after step t-1 as the reference run had it
                col = lambda x: torch.as_tensor(np.tile(np.asarray(x, np.float64).reshape(-1, 1), (1, n)))
                env.load_state_dict(dict(pos=col(case["pos"][t - 1]), rot=col(case["rot"][t - 1]), vel=col(case["vel"][t - 1]),
                                         omega=col(case["omega"][t - 1]), q=col(case["q"][t - 1]), qd=col(case["qd"][t - 1]),
                                         potential=torch.full((n,), float(case["potential"][t - 1]), dtype=torch.float64),
                                         feet_contact=col(case["feet_contact"][t - 1]).float(),
                                         steps=torch.full((n,), int(case["steps"][t - 1]), dtype=torch.int32)))
            a = torch.as_tensor(np.tile(case["actions"][t], (n, 1)))
            obs, rew, done, info = env.step(a)
            obs, r5 = obs.cpu().numpy(), info["rewards"].cpu().numpy()
            assert np.array_equal(obs[0], obs[1]) and np.array_equal(obs[0], obs[2])
            worst["obs"] = max(worst["obs"], float(np.max(np.abs(obs[0] - case["obs"][t]))))
            for name, val in (("pos", env.pos), ("q", env.q), ("qd", env.qd), ("vel", env.vel)):
                worst["state"] = max(worst["state"], float(np.max(np.abs(val[:, 0].cpu().numpy() - case[name][t]))))
            assert np.max(np.abs(obs[0] - case["obs"][t])) <= 1e-6, (t, np.max(np.abs(obs[0] - case["obs"][t])))
            assert np.allclose(r5[0], case["rewards"][t], rtol=1e-5, atol=1e-5), (t, r5[0], case["rewards"][t])
            assert abs(float(rew[0]) - case["reward"][t]) <= 1e-5 * max(1.0, abs(case["reward"][t])), t
            assert bool(done[0]) == bool(case["done"][t]) and int(info["steps"][0]) == int(case["steps"][t]), t
            assert np.array_equal(env.feet_contact[:, 0].cpu().numpy(), case["feet_contact"][t]), t
            t += 1
    assert worst["state"] <= 1e-6, worst              # round-off between the two implementations of the engine, 50 free steps
    print(task, mapping, worst)


def test_free_flight_invariants_on_gpu():
    """No gravity-free mode exists in the product API, so: in free fall (high above the ground) the
    total momentum of every env changes by exactly m*g*t and the joints keep moving smoothly."""
    m = copy.deepcopy(MODELS["humanoid@mujoco"])          # (no body damping: momentum is conserved)
    m.joint_damping[:] = 0
    m.joint_stiffness[:] = 0
    m.joint_lo[:] = -100
    m.joint_hi[:] = 100
    n = 64
    env = _make("MetaHumanoidEnv", [m], n)
    env.reset(seed=1)
    sd = env.state_dict()
    sd["pos"][2] += 50.0
    rs = np.random.RandomState(2)
    sd["qd"] = torch.as_tensor(rs.uniform(-2, 2, (17, n)))
    sd["omega"] = torch.as_tensor(rs.uniform(-1, 1, (3, n)))
    env.load_state_dict(sd)

    def momentum(e):
        s = abd.State(m)
        s.pos, s.rot = env.pos[:, e].cpu().numpy(), env.rot[:, e].cpu().numpy().reshape(3, 3)
        s.v, s.w = env.vel[:, e].cpu().numpy(), env.omega[:, e].cpu().numpy()
        s.q, s.qd = env.q[:, e].cpu().numpy(), env.qd[:, e].cpu().numpy()
        return abd.momentum(m, s)[0], sum(abd.energy(m, s))

    before = [momentum(e) for e in range(0, n, 16)]
    for _ in range(10):
        env.step(torch.zeros(n, 17))
    after = [momentum(e) for e in range(0, n, 16)]
    mg_t = -9.8 * m.body_mass.sum() * 10 * 0.02
    for (p0, e0), (p1, e1) in zip(before, after):
        assert abs((p1 - p0)[2] - mg_t) < 0.02 * abs(mg_t)
        assert np.abs((p1 - p0)[:2]).max() < 0.02 * abs(mg_t)
        assert abs(e1 - e0) < 0.02 * abs(e0)                 # energy conserved in free fall (first-order drift)


def test_full_size_properties_8192():
    """BASELINE config C4 size: 8 192 humanoids over 3 body variants; determinism, env-permutation
    equivariance, finite outputs, nobody sinks below the floor, masked reset touches only its envs."""
    models = [MODELS[k] for k in ("humanoid", "humanoid_tra_000", "humanoid_tra_137")]
    n = 8192
    ids = torch.arange(n, dtype=torch.int32) % 3
    perm = torch.randperm(n)
    a = _make("MetaHumanoidEnv", models, n, ids)
    b = _make("MetaHumanoidEnv", models, n, ids[perm])
    noise = np.random.RandomState(3).uniform(-0.1, 0.1, (n, 17))
    oa = a.reset(joint_noise=noise)
    ob = b.reset(joint_noise=noise[perm.numpy()])
    assert torch.equal(oa[perm.cuda()], ob)
    g = torch.Generator().manual_seed(0)
    for t in range(12):
        act = torch.rand(n, 17, generator=g) * 2 - 1
        oa, ra, da, ia = a.step(act)
        ob, rb, db, ib = b.step(act[perm])
        assert torch.equal(oa[perm.cuda()], ob) and torch.equal(ra[perm.cuda()], rb) and torch.equal(da[perm.cuda()], db)
    assert torch.isfinite(oa).all() and torch.isfinite(ra).all()
    assert oa.shape == (n, 44)
    assert float(a.pos[2].min()) > 0.0                     # torso stays above the floor
    before = a.state_dict()
    mask = torch.zeros(n, dtype=torch.bool)
    mask[::5] = True
    a.reset(mask=mask, seed=9)
    after = a.state_dict()
    keep = ~mask.cuda()
    for k in ("pos", "q", "qd", "vel"):
        assert torch.equal(before[k][:, keep], after[k][:, keep]), k
    assert (after["steps"][mask.cuda()] == 0).all() and (after["qd"][:, mask.cuda()] == 0).all()


def _sample_envs(n, groups_wanted=8):
    """Aligned groups of 8 envs: the first and the last of the batch and groups covering the env -> XCD rotations
    mg::env_of_block applies (metagym_amd/csrc/mg_common.h)."""
    rot = lambda q: (q + (q >> 3) + (q >> 6) + (q >> 9)) & 7
    groups, seen, q = [0, n // 8 - 1], {rot(0), rot(n // 8 - 1)}, 37
    while len(groups) < groups_wanted:
        if rot(q) not in seen or len(seen) == 8:
            groups.append(q)
            seen.add(rot(q))
        q = (q + 263) % (n // 8)
    return np.asarray(sorted(8 * g + x for g in groups for x in range(8)))


def test_c4_timed_configuration_sampled_against_oracle():
    """BASELINE configs[3] exactly as bench.py times it (secondary_workloads: 8 192 humanoids, env e runs
    humanoid_var_tra_<e mod 256>.xml, the envs' default preset, self-collision on, U(-1, 1) actions, fused auto-reset,
    max_steps 1000, a batch rolled to its steady state first): 64 sampled envs replayed on oracle/walker_oracle.c (the
    engine's scalar C restatement, pinned to oracle/abd.py by tests/test_oracle_walker_c.py), state 1e-7, float32 observation
    2e-5, reward terms, done and feet flags.
      phase A  the first 10 env steps after reset. Every variant starts with its feet 9 cm in the ground
               (gen_variant_humanoids.py:44 lowers the torso by 0.20): the contact ERP of 0.9 (scene_bases.py:55) turns that into
               a separation velocity of ~17 m/s in the first sub-step and the robot is airborne afterwards;
      phase B  after 200 more (unchecked) steps, when the batch is a mix of robots in flight, landing, tumbling and restarting,
               the oracle takes over 32 of the sampled envs and the 32 envs whose torso is lowest at that moment, and follows
               12 env steps (handed the GPU state again every 4) — ground
               contacts, joint limits, self-contacts, and the in-launch restarts of envs whose episode ends (the oracle resets
               with the joint noise the kernel drew)."""
    import ctypes as C
    from metagym_amd.metalocomotion import MetaHumanoidEnv, variants
    from oracle import walker_c
    n = 8192
    models = variants.models("humanoid", "TRAIN")                     # == bench.py's C4 input
    assert len(models) == 256 and str(models[0].preset) == "bullet"
    env = MetaHumanoidEnv(num_envs=n, device="cuda:0", auto_reset=True, max_steps=1000, seed=3)
    env.set_task(models)
    assert env.body_damping == (0.04, 0.04) and env.max_coordinate_velocity == 100.0
    ids = env.task_id.cpu().numpy()
    assert np.array_equal(ids, np.arange(n) % 256)
    rs = np.random.RandomState(4)
    noise = rs.uniform(-0.1, 0.1, (n, 17))
    obs = env.reset(joint_noise=noise).cpu().numpy()
    sample = _sample_envs(n)
    lib = walker_c.load()
    power = abd.HUMANOID_MOTOR_POWER * 0.41
    dptr, fptr = (lambda a: a.ctypes.data_as(C.POINTER(C.c_double))), (lambda a: a.ctypes.data_as(C.POINTER(C.c_float)))
    cenvs = {}

    def oracle_env(e):
        m = models[ids[e]]
        cm, table = walker_c.make_model(m, power)
        prm = walker_c.humanoid_params(m, floor_in_parts=0, max_steps=1000)      # an env's first reset (walker_base_env.py:30-31)
        assert prm.body_linear_damping == 0.04 and prm.max_coordinate_velocity == 100.0
        ce = walker_c.Env()
        o = np.zeros(44, np.float32)
        lib.wo_env_reset(C.byref(cm), C.byref(prm), C.byref(ce), dptr(np.ascontiguousarray(noise[e])), fptr(o))
        ce.floor_known = 1
        prm.floor_in_parts = 1
        cenvs[int(e)] = (cm, table, prm, ce)
        return o
    for e in sample:
        assert np.allclose(obs[e], oracle_env(e), rtol=0, atol=1e-6), e
    worst = dict(state=0.0, obs=0.0)
    count = dict(contacts=0, ends=0, zmax=0.0)

    def hand_over():
        """The oracle takes over the sampled envs as they are on the GPU now."""
        st = {k: getattr(env, k).cpu().numpy() for k in ("pos", "rot", "vel", "omega", "q", "qd", "feet_contact", "steps", "potential")}
        for e in sample:
            ce = cenvs[int(e)][3]
            ce.s.pos[:], ce.s.rot[:], ce.s.vel[:], ce.s.omega[:] = list(st["pos"][:, e]), list(st["rot"][:, e]), list(st["vel"][:, e]), list(st["omega"][:, e])
            ce.s.q[:17], ce.s.qd[:17] = list(st["q"][:, e]), list(st["qd"][:, e])
            ce.potential, ce.steps, ce.floor_known, ce.initial_z_unset = float(st["potential"][e]), int(st["steps"][e]), 1, 0
            ce.feet_contact[:2] = [float(x) for x in st["feet_contact"][:, e]]
        return st

    def compare_steps(n_steps, phase, resync=0):
        for t in range(n_steps):
            if resync and t and t % resync == 0:
                hand_over()
            a = rs.uniform(-1.0, 1.0, (n, 17)).astype(np.float32)
            obs, rew, done, info = env.step(torch.as_tensor(a))
            obs, rew, done, r5 = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy(), info["rewards"].cpu().numpy()
            st = {k: getattr(env, k).cpu().numpy() for k in ("pos", "rot", "vel", "omega", "q", "qd", "feet_contact", "steps")}
            for e in sample:
                cm, table, prm, ce = cenvs[int(e)]
                o = np.zeros(44, np.float32)
                r, r5c = C.c_double(), (C.c_double * 5)()
                d = lib.wo_env_step(C.byref(cm), C.byref(prm), C.byref(ce), fptr(np.ascontiguousarray(a[e])), fptr(o), C.byref(r), r5c)
                assert bool(done[e]) == bool(d), (phase, t, e)
                assert np.allclose(r5[e], list(r5c), rtol=1e-5, atol=1e-4) and abs(rew[e] - r.value) < 1e-4 * max(1.0, abs(r.value)), (phase, t, e)
                if d:       # restarted inside the launch: same reset on the oracle, with the joint noise the kernel drew
                    count["ends"] += 1
                    lib.wo_env_reset(C.byref(cm), C.byref(prm), C.byref(ce), dptr(np.ascontiguousarray(st["q"][:, e])), fptr(o))
                cs = ce.s
                err = max(np.abs(st["pos"][:, e] - np.array(cs.pos[:])).max(), np.abs(st["rot"][:, e] - np.array(cs.rot[:])).max(),
                          np.abs(st["q"][:, e] - np.array(cs.q[:17])).max(),
                          1e-2 * np.abs(st["vel"][:, e] - np.array(cs.vel[:])).max(), 1e-2 * np.abs(st["omega"][:, e] - np.array(cs.omega[:])).max(),
                          1e-2 * np.abs(st["qd"][:, e] - np.array(cs.qd[:17])).max())
                worst["state"], worst["obs"] = max(worst["state"], err), max(worst["obs"], float(np.abs(obs[e] - o).max()))
                assert err < 1e-7, (phase, t, e, err)
                assert np.allclose(obs[e], o, rtol=0, atol=2e-5), (phase, t, e, np.abs(obs[e] - o).max())
                assert np.array_equal(st["feet_contact"][:, e], np.array(ce.feet_contact[:2])) and st["steps"][e] == ce.steps, (phase, t, e)
                count["contacts"] += int(st["feet_contact"][:, e].sum())
                count["zmax"] = max(count["zmax"], float(cs.pos[2]))
            assert np.isfinite(obs).all()

    compare_steps(10, "A")
    for t in range(200):                                              # to the steady state (unchecked)
        env.step(torch.as_tensor(rs.uniform(-1.0, 1.0, (n, 17)).astype(np.float32)))
    # phase B follows the fixed sample's first half plus the 32 envs whose torso is lowest right now: the batch's steady state
    # is mostly flight (a restart throws the robot up again), the compared steps should be the ones on the ground
    z = env.pos[2].cpu().numpy()
    lowest = [int(e) for e in np.argsort(z) if int(e) not in set(int(x) for x in sample[:32])][:32]
    sample = np.asarray(sorted([int(e) for e in sample[:32]] + lowest))
    for e in lowest:
        oracle_env(e)
    st = hand_over()
    low = int((st["pos"][2, sample] < 1.0).sum())
    before = dict(count)
    compare_steps(12, "B", resync=4)          # (contact-rich tumbling amplifies round-off: 2e-7 after 9 free-running steps)
    print("C4: %d sampled envs, 10 + 12 env steps, max |state diff| GPU vs C oracle %.2e, obs %.2e; phase B: %d of the sampled torsos "
          "below 1 m at hand-over, %d foot-contact flags, %d episode ends; highest torso %.2f m"
          % (len(sample), worst["state"], worst["obs"], low, count["contacts"] - before["contacts"], count["ends"] - before["ends"], count["zmax"]))
    # Ground contact inside the compared steps shows as episode ENDS (a torso below 0.5 m, walker_base_env.py:47-51) followed by the
    # in-launch restart, which the oracle follows too. (The batch is airborne most of the time: the contact ERP of 0.9 returns a
    # landing at 20 m/s as a take-off at 18 — so few torsos are low at any instant and the feet flags are reported, not counted on.)
    assert count["ends"] - before["ends"] >= 4


def test_c4_grounded_configuration_sampled_against_oracle():
    """The contact-rich sibling of the test above — bench.py's `C4_grounded_*` exactly as it is timed: the same
    8 192 humanoids over the 256 TRAIN variants, default preset, self-collision on, but every robot starts STANDING on the floor
    (mjcf.grounded: the base lifted by the 9.3 cm the reference's variants start inside it), actions U(-0.1, 0.1), no auto-reset —
    so the batch sags, kneels, falls and then LIES on the ground with many proxies down, the regime where ground-contact rows,
    friction pyramids, joint-limit rows and the PGS sweeps carry load (today's C4 is airborne 99 % of the time). 64 sampled envs
    on oracle/walker_oracle.c: state 1e-7, observation 2e-5, the five reward terms (joints_at_limit_cost, walker_base_env.py:68 —
    identically zero in flight), done, the feet flags (walker_base_env.py:57-63).
      phase A  40 env steps from the reset: standing on two feet, the legs folding, the knees and hands reaching the floor;
      phase B  150 unchecked steps later — everybody down — 12 more.
    The oracle is handed the GPU state every 4 steps in both phases (a contact row exists only while a proxy penetrates, so
    round-off decides on which sub-step a resting contact flickers; free-running copies drift apart like the tumbling ones above).
    The instrumented build of the oracle reports the solver's load over the compared steps: constraint rows per sub-step."""
    import ctypes as C
    from metagym_amd.metalocomotion import MetaHumanoidEnv, mjcf, variants
    from oracle import walker_c
    n = 8192
    models = [mjcf.grounded(m) for m in variants.models("humanoid", "TRAIN")]      # == bench.py's C4_grounded input
    assert all(abs(mjcf.rest_lowest_point(m)) < 1e-12 for m in models[:8])
    env = MetaHumanoidEnv(num_envs=n, device="cuda:0", auto_reset=False, max_steps=1000, seed=3)
    env.set_task(models)
    ids = env.task_id.cpu().numpy()
    rs = np.random.RandomState(14)
    noise = rs.uniform(-0.1, 0.1, (n, 17))
    obs = env.reset(joint_noise=noise).cpu().numpy()
    sample = _sample_envs(n)
    lib = walker_c.load(count_flops=True)                             # the same C, plus the row / contact counters
    power = abd.HUMANOID_MOTOR_POWER * 0.41
    dptr, fptr = (lambda a: a.ctypes.data_as(C.POINTER(C.c_double))), (lambda a: a.ctypes.data_as(C.POINTER(C.c_float)))
    cenvs = {}
    for e in sample:
        cm, table = walker_c.make_model(models[ids[e]], power)
        prm = walker_c.humanoid_params(models[ids[e]], floor_in_parts=0, max_steps=1000)
        ce = walker_c.Env()
        o = np.zeros(44, np.float32)
        lib.wo_env_reset(C.byref(cm), C.byref(prm), C.byref(ce), dptr(np.ascontiguousarray(noise[e])), fptr(o))
        ce.floor_known, prm.floor_in_parts = 1, 1
        cenvs[int(e)] = (cm, table, prm, ce)
        assert np.allclose(obs[e], o, rtol=0, atol=1e-6), e
    keys = ("pos", "rot", "vel", "omega", "q", "qd", "feet_contact", "steps", "potential")

    def hand_over():
        st = {k: getattr(env, k).cpu().numpy() for k in keys}
        for e in sample:
            ce = cenvs[int(e)][3]
            ce.s.pos[:], ce.s.rot[:], ce.s.vel[:], ce.s.omega[:] = list(st["pos"][:, e]), list(st["rot"][:, e]), list(st["vel"][:, e]), list(st["omega"][:, e])
            ce.s.q[:17], ce.s.qd[:17] = list(st["q"][:, e]), list(st["qd"][:, e])
            ce.potential, ce.steps, ce.floor_known, ce.initial_z_unset = float(st["potential"][e]), int(st["steps"][e]), 1, 0
            ce.feet_contact[:2] = [float(x) for x in st["feet_contact"][:, e]]
        return st
    worst = dict(state=0.0, obs=0.0)
    seen = dict(feet=0, limit_steps=0, done=0)

    def compare_steps(n_steps, phase):
        counters = (C.c_ulonglong * 9)()
        lib.wo_flops_read(counters, 1)
        for t in range(n_steps):
            if t and t % 4 == 0:
                hand_over()
            a = rs.uniform(-0.1, 0.1, (n, 17)).astype(np.float32)
            obs, rew, done, info = env.step(torch.as_tensor(a))
            obs, rew, done, r5 = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy(), info["rewards"].cpu().numpy()
            st = {k: getattr(env, k).cpu().numpy() for k in keys}
            for e in sample:
                cm, table, prm, ce = cenvs[int(e)]
                o = np.zeros(44, np.float32)
                r, r5c = C.c_double(), (C.c_double * 5)()
                d = lib.wo_env_step(C.byref(cm), C.byref(prm), C.byref(ce), fptr(np.ascontiguousarray(a[e])), fptr(o), C.byref(r), r5c)
                assert bool(done[e]) == bool(d), (phase, t, e)
                assert np.allclose(r5[e], list(r5c), rtol=1e-5, atol=1e-4) and abs(rew[e] - r.value) < 1e-4 * max(1.0, abs(r.value)), (phase, t, e)
                cs = ce.s
                err = max(np.abs(st["pos"][:, e] - np.array(cs.pos[:])).max(), np.abs(st["rot"][:, e] - np.array(cs.rot[:])).max(),
                          np.abs(st["q"][:, e] - np.array(cs.q[:17])).max(),
                          1e-2 * np.abs(st["vel"][:, e] - np.array(cs.vel[:])).max(), 1e-2 * np.abs(st["omega"][:, e] - np.array(cs.omega[:])).max(),
                          1e-2 * np.abs(st["qd"][:, e] - np.array(cs.qd[:17])).max())
                worst["state"], worst["obs"] = max(worst["state"], err), max(worst["obs"], float(np.abs(obs[e] - o).max()))
                assert err < 1e-7, (phase, t, e, err)
                assert np.allclose(obs[e], o, rtol=0, atol=2e-5), (phase, t, e, np.abs(obs[e] - o).max())
                assert np.array_equal(st["feet_contact"][:, e], np.array(ce.feet_contact[:2])) and st["steps"][e] == ce.steps, (phase, t, e)
                seen["feet"] += int(st["feet_contact"][:, e].sum())
                seen["limit_steps"] += int(r5c[3] < 0.0)
                seen["done"] += int(d)
            assert np.isfinite(obs).all()
        assert lib.wo_flops_read(counters, 1) == 1
        return counters[7] / float(counters[6]), counters[8] / float(counters[6])

    rows_a, cont_a = compare_steps(40, "A")
    for t in range(150):
        env.step(torch.as_tensor(rs.uniform(-0.1, 0.1, (n, 17)).astype(np.float32)))
    st = hand_over()
    z = st["pos"][2]
    rows_b, cont_b = compare_steps(12, "B")
    ground = float((env.feet_contact.sum(0) > 0).float().mean())
    print("C4 grounded: %d sampled envs, 40 + 12 env steps, max |state diff| GPU vs C oracle %.2e, obs %.2e; constraint rows per sub-step "
          "%.1f (contacts %.2f) while standing / falling, %.1f (contacts %.2f) lying; torso heights at hand-over %.2f .. %.2f m (batch "
          "median %.2f); %d foot flags, %d env steps with joints at their limits, %d done flags; %.0f %% of the batch with a foot on the "
          "ground at the end" % (len(sample), worst["state"], worst["obs"], rows_a, cont_a, rows_b, cont_b, float(z[sample].min()),
                                 float(z[sample].max()), float(np.median(z)), seen["feet"], seen["limit_steps"], seen["done"], 100 * ground))
    assert float(np.median(z)) < 0.6                                   # the batch IS on the ground
    assert rows_b > 8.0 and cont_b > 2.0                               # ... and the solver carries load (C4 in flight: 3.6 rows, 0.2 contacts)
    assert seen["feet"] > 200 and seen["limit_steps"] > 200 and seen["done"] > 200


@pytest.mark.parametrize("mapping", ["wave", "lane"])
@pytest.mark.parametrize("margin", ["relative", 0.02, 0.0])
def test_contact_margin_and_deepest_first_cap_match_oracle(mapping, margin):
    """mg_walker_params.contact_margin (ABI 7) and the solver's contact cap, both mappings against oracle/abd.py: humanoids laid
    on their backs a little inside / on / above the floor — more than 12 contact candidates for most of them (the cap keeps the 12
    DEEPEST, in candidate order), speculative rows for the hovering ones, feet flags from every proxy inside the margin. State to
    1e-7 over 8 env steps, feet flags exactly; the oracle's own bookkeeping proves the cases were met."""
    names = ["humanoid", "humanoid_tra_137"]
    models = [MODELS[k] for k in names]
    n = 8
    env = _make("MetaHumanoidEnv", models, n, max_steps=1000, mapping=mapping, contact_margin=margin, contact_erp=0.1)   # (soft push-out: the pile-up lasts)
    from metagym_amd.metalocomotion.mjcf import contact_margins
    assert env.contact_margin == margin and env._params_c.contact_margin == (0.0 if margin == "relative" else margin)
    assert env._params_c.sphere_margin_in_table == int(margin == "relative")       # Bullet's per-link rule: the margins ride in the model rows
    ids = env.task_id.cpu().numpy()
    nj = env.n_joints
    rs = np.random.RandomState(4)
    noise = rs.uniform(-0.1, 0.1, (n, nj))
    env.reset(joint_noise=noise)
    Ry = abd.rodrigues(np.array([0.0, 1.0, 0.0]), -np.pi / 2)
    lift = np.array([-0.08, -0.08, -0.10, -0.10, -0.03, 0.005, 0.012, 0.018])     # lowest proxy surface: inside ... above the floor
    oenvs = []
    for e in range(n):
        m = models[ids[e]]
        kw = world_kw(m)
        kw["contact_margin"] = contact_margins(m, margin)
        o = abd.WalkerEnv(m, prm=abd.Params(friction=0.8 * float(m.geom_friction), self_friction=float(m.geom_friction) ** 2, erp=0.1, **kw),
                          max_steps=1000)
        o.reset(noise[e])
        o.s.rot = Ry @ o.s.rot
        o.s.pos[2] = 0.0
        kin = abd.kinematics(m, o.s)
        o.s.pos[2] = lift[e] - min((kin["o"][b] + kin["R"][b] @ m.sph_pos[g])[2] - m.sph_radius[g] for g, b in enumerate(m.sph_body))
        oenvs.append(o)
    env.rot.copy_(torch.as_tensor(np.stack([o.s.rot.reshape(9) for o in oenvs], 1), device=env.rot.device))
    env.pos.copy_(torch.as_tensor(np.stack([o.s.pos for o in oenvs], 1), device=env.pos.device))
    over_cap, speculative, worst = 0, 0, 0.0
    for t in range(8):
        a = rs.uniform(-0.3, 0.3, (n, nj)).astype(np.float32)
        for e in range(n):                                  # what the oracle is about to see in the first sub-step of this step
            kin = abd.kinematics(models[ids[e]], oenvs[e].s)
            cands, _ = abd.contact_candidates(models[ids[e]], oenvs[e].s, kin, oenvs[e].prm)
            over_cap += int(len(cands) > 12)
            speculative += int(any(c["depth"] < 0.0 for c in cands))
        env.step(torch.as_tensor(a))
        q, pos, fc = env.q.cpu().numpy().T, env.pos.cpu().numpy().T, env.feet_contact.cpu().numpy().T
        for e in range(n):
            oenvs[e].step(a[e])
            s = oenvs[e].s
            worst = max(worst, np.abs(q[e] - s.q).max(), np.abs(pos[e] - s.pos).max())
            assert np.allclose(q[e], s.q, rtol=0, atol=1e-7), (t, e, np.abs(q[e] - s.q).max())
            assert np.allclose(pos[e], s.pos, rtol=0, atol=1e-7), (t, e)
            assert np.array_equal(fc[e], oenvs[e].feet_contact), (t, e)
    assert over_cap >= 4, over_cap
    assert (speculative >= 8) if margin != 0.0 else (speculative == 0)
    print(mapping, "margin", margin, "max |state diff| %.2e; env-steps starting with > 12 candidates: %d, with speculative ones: %d"
          % (worst, over_cap, speculative))


@pytest.mark.parametrize("mapping", ["wave", "lane"])
def test_self_collision_matches_oracle(mapping):
    """Legs swung into each other (and an arm into the torso side) in mid-air: the capsule-capsule
    self-contact rows (robot_bases.py:119 flags) must act exactly like the oracle's — and switching
    them off must change the outcome (the scenario really exercises them)."""
    m = MODELS["humanoid"]
    n = 4
    outcomes = {}
    for self_on in (True, False):
        env = _make("MetaHumanoidEnv", [m], n, mapping=mapping, self_collision=self_on)
        env.reset(joint_noise=np.zeros((n, 17)))
        sd = env.state_dict()
        sd["pos"][2] += 5.0
        qd = np.zeros((17, n))
        for e in range(n):
            qd[m.joint_names.index("right_hip_x"), e] = 1.5 + 0.5 * e
            qd[m.joint_names.index("left_hip_x"), e] = 1.5 + 0.5 * e
            qd[m.joint_names.index("right_shoulder1"), e] = 1.0 * e
        sd["qd"] = torch.as_tensor(qd)
        env.load_state_dict(sd)
        oenvs = []
        for e in range(n):
            oe = _oracle_env(m)
            oe.prm.self_collision = self_on
            oe.reset(np.zeros(17))
            oe.s.pos[2] += 5.0
            oe.s.qd = qd[:, e].copy()
            oenvs.append(oe)
        saw_self = False
        for t in range(6):
            env.step(torch.zeros(n, 17))
            q = env.q.cpu().numpy().T
            for e in range(n):
                for _ in range(4):
                    kin = abd.kinematics(m, oenvs[e].s)
                    saw_self = saw_self or any(r[4] == -2 for r in abd.constraint_rows(m, oenvs[e].s, kin, oenvs[e].prm))
                    abd.substep(m, oenvs[e].s, np.zeros(17), oenvs[e].prm)
                assert np.allclose(q[e], oenvs[e].s.q, rtol=0, atol=1e-7), (self_on, t, e, np.abs(q[e] - oenvs[e].s.q).max())
        outcomes[self_on] = env.q.cpu().numpy().copy()
        assert saw_self == self_on
    assert np.abs(outcomes[True] - outcomes[False]).max() > 1e-2


@pytest.mark.parametrize("mapping", ["wave", "lane"])
def test_auto_reset_equals_explicit_masked_reset(mapping):
    """Fused auto-reset: an env that ends its episode restarts inside the launch. Against a
    twin env driven with explicit `reset(mask=done, joint_noise=<the noise the kernel drew>)`: identical
    state and observations afterwards; the drawn joint angles are U(-0.1, 0.1); reward/done of the
    ending step are unaffected; noise is keyed by global env id (shard invariance)."""
    models = [MODELS[k] for k in ("humanoid", "humanoid_tra_000")]
    n, T = 24, 12
    auto = _make("MetaHumanoidEnv", models, n, max_steps=5, mapping=mapping, auto_reset=True, seed=11)
    twin = _make("MetaHumanoidEnv", models, n, max_steps=5, mapping=mapping)
    rs = np.random.RandomState(3)
    noise = rs.uniform(-0.1, 0.1, (n, auto.n_joints))
    o0, o1 = auto.reset(joint_noise=noise), twin.reset(joint_noise=noise)
    assert torch.equal(o0, o1)
    ended_total = 0
    for t in range(T):
        a = torch.as_tensor(rs.uniform(-1, 1, (n, auto.n_joints)).astype(np.float32)).cuda()
        oa, ra, da, _ = auto.step(a)
        ob, rb, db, _ = twin.step(a)
        assert torch.equal(ra, rb) and torch.equal(da, db)
        d = db.clone()
        if bool(d.any()):
            ended_total += int(d.sum())
            q_new = auto.q.T[d].cpu().numpy()                  # the joint noise the kernel drew
            assert (np.abs(q_new) <= 0.1).all() and np.abs(q_new).max() > 0.01
            jn = auto.q.T.cpu().numpy().copy()
            ob = twin.reset(mask=d, joint_noise=jn).clone()
        else:
            ob = ob.clone()
        assert torch.equal(oa[d], ob[d])                       # first obs of the new episode
        assert torch.equal(oa[~d], ob[~d])
        sa, sb = auto.state_dict(), twin.state_dict()
        for k in sa:
            if not torch.is_tensor(sa[k]):
                continue                           # global_step / np_random: host-side stream positions
            # the fused reset runs the wave kernel's observation code, the explicit one the lane-per-env reset
            # kernel's; with FMA contraction on in this engine they may differ in the last bit of a double
            if sa[k].dtype == torch.float64:
                assert torch.allclose(sa[k], sb[k], rtol=1e-12, atol=1e-12), (t, k)
            else:
                assert torch.equal(sa[k], sb[k]), (t, k)
        twin.load_state_dict({k: v for k, v in sa.items() if torch.is_tensor(v)})          # keep the twins on identical bits so the comparison stays sharp
    assert ended_total >= n                                    # max_steps=5: everyone restarted at least once
    # shard invariance: envs 12..23 run as their own shard draw the same noise
    full = _make("MetaHumanoidEnv", models, n, max_steps=2, mapping=mapping, auto_reset=True, seed=5)
    half = _make("MetaHumanoidEnv", models, n // 2, task_ids=full.task_id[n // 2:].cpu(), max_steps=2, mapping=mapping,
                 auto_reset=True, seed=5, env_id_base=n // 2)
    full.reset(joint_noise=noise)
    half.reset(joint_noise=noise[n // 2:])
    for t in range(3):
        a = torch.as_tensor(rs.uniform(-1, 1, (n, full.n_joints)).astype(np.float32)).cuda()
        of, _, df, _ = full.step(a)
        oh, _, dh, _ = half.step(a[n // 2:])
        assert torch.equal(of[n // 2:], oh) and torch.equal(df[n // 2:], dh)
    assert torch.equal(full.q[:, n // 2:], half.q)


def test_in_launch_raw_torque_actuation_equals_the_action_path():
    """mg_walker_params.actuation = 2 (raw float64 torques, re-applied before every sub-step inside the launch) fed
    gain * clip(a) is the action path itself: same states bit for bit over 6 env steps of the ant (whose apply_action
    multiplies in float64, walker_base.py:26-29) — which ties the in-launch actuators to the oracle-checked engine path.
    And the sub-step log of those launches ends on the state arrays."""
    models = [MODELS["ant"]]
    n = 96
    a_env, b_env = _make("MetaAntEnv", models, n), _make("MetaAntEnv", models, n)
    rs = np.random.RandomState(5)
    noise = rs.uniform(-0.1, 0.1, (n, 8))
    a_env.reset(joint_noise=noise); b_env.reset(joint_noise=noise)
    gain = 100.0 * a_env.power                                             # motor_power None -> 100 (robot_bases.py:93)
    log = torch.empty(a_env.frame_skip, 3 * 8 + 7, n, dtype=torch.float64, device="cuda:0")
    for k in range(6):
        act = torch.as_tensor(rs.uniform(-1.3, 1.3, (n, 8)).astype(np.float32))
        a_env.step(act)
        torque = (gain * act.clamp(-1.0, 1.0).double()).t().contiguous().cuda()
        b_env.step_actuated(torque, raw_torque=True, log=log)
        for key in ("pos", "rot", "vel", "omega", "q", "qd"):
            assert torch.equal(getattr(a_env, key), getattr(b_env, key)), (k, key)
        assert torch.equal(a_env._obs, b_env._obs) and torch.equal(a_env._reward, b_env._reward)
        assert torch.equal(log[-1, 0:8], b_env.q) and torch.equal(log[-1, 8:16], b_env.qd) and torch.equal(log[-1, 16:24], torque)
        quat = log[-1, 24:28]
        assert torch.allclose((quat * quat).sum(0), torch.ones(n, dtype=torch.float64, device="cuda:0"), atol=1e-12)


def _terrain_for_tests():
    """A little course around the origin: a platform, a two-step stair, an up-ramp and a steep low-friction down-ramp —
    (half_extents, position, quaternion, friction) like metagym_amd.quadrupedal.terrain's boxes."""
    def yq(a):
        return [0.0, np.sin(a / 2), 0.0, np.cos(a / 2)]
    return [([0.6, 2.5, 0.05], [0.0, 0.0, 0.05], yq(0.0), 1.0),            # platform, top at z = 0.1
            ([0.15, 2.5, 0.05], [0.75, 0.0, 0.15], yq(0.0), 5.0),          # step 1, top 0.2
            ([0.15, 2.5, 0.05], [1.05, 0.0, 0.25], yq(0.0), 5.0),          # step 2, top 0.3
            ([0.8, 2.5, 0.01], [-1.3, 0.0, 0.25], yq(0.25), 0.9),          # ramp rising towards -x
            ([0.5, 2.5, 0.01], [0.0, 3.2, 0.3], yq(-0.5), 0.2)]            # off to the side, slippery


def _oracle_boxes(spec, geom_friction):
    out = []
    for half, pos, (x, y, z, w), mu in spec:
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        out.append((np.array(pos, float), R, np.array(half, float), mu * geom_friction))
    return out


@pytest.mark.parametrize("robot", ["humanoid", "ant"])
def test_terrain_boxes_match_oracle_trajectory(robot):
    """Static terrain boxes (mg_walker_params.terrain): robots dropped at different places of a small course — platform,
    stair edges, ramps, next to it on the plain ground — follow the oracle's trajectories (same deepest-box contact rule,
    per-box friction), and the terrain actually carries them."""
    ant = robot == "ant"
    names = ["ant", "ant_tra_005"] if ant else ["humanoid", "humanoid_tra_137"]
    models = [MODELS[k] for k in names]
    n = 5 * len(models)
    spec = _terrain_for_tests()
    env = _make("MetaAntEnv" if ant else "MetaHumanoidEnv", models, n, max_steps=1000)
    env.set_terrain(spec)
    ids = env.task_id.cpu().numpy()
    nj = env.n_joints
    rs = np.random.RandomState(3)
    noise = rs.uniform(-0.1, 0.1, (n, nj))
    env.reset(joint_noise=noise)
    starts = np.array([[0.0, 0.0, 0.12], [0.8, 0.1, 0.32], [-1.3, -0.2, 0.35], [0.45, 0.0, 0.2], [3.0, 0.0, 0.0]])   # last: off the course
    offs = starts[np.arange(n) % len(starts)]
    env.pos += torch.as_tensor(offs.T, device=env.pos.device)
    oenvs = []
    for e in range(n):
        m = models[ids[e]]
        kw = dict(friction=0.8 * float(m.geom_friction), self_friction=float(m.geom_friction) ** 2,
                  terrain=_oracle_boxes(spec, float(m.geom_friction)), **world_kw(m))
        if ant:
            o = abd.WalkerEnv(m, prm=abd.Params(power=2.5, **kw), motor_power=np.full(nj, 100.0), alive_z=0.26, alive_bonus=1.0,
                              initial_z=None, torque_f32=False, max_steps=1000)
        else:
            o = abd.WalkerEnv(m, prm=abd.Params(**kw), max_steps=1000)
        o.reset(noise[e])
        o.s.pos += offs[e]
        oenvs.append(o)
    worst, on_terrain = 0.0, 0
    for t in range(30):
        a = rs.uniform(-0.6, 0.6, (n, nj)).astype(np.float32)
        env.step(torch.as_tensor(a))
        q, pos, fc = env.q.cpu().numpy().T, env.pos.cpu().numpy().T, env.feet_contact.cpu().numpy().T
        for e in range(n):
            oenvs[e].step(a[e])
            s = oenvs[e].s
            worst = max(worst, np.abs(q[e] - s.q).max(), np.abs(pos[e] - s.pos).max())
            assert np.allclose(q[e], s.q, rtol=0, atol=2e-7), (t, e, np.abs(q[e] - s.q).max())
            assert np.allclose(pos[e], s.pos, rtol=0, atol=2e-7), (t, e)
            assert np.array_equal(fc[e], oenvs[e].feet_contact), (t, e)
    for e in range(n):      # whoever started over the course is still above the plain ground level it would have fallen to
        kin = abd.kinematics(models[ids[e]], oenvs[e].s)
        lowest = min((kin["o"][b] + kin["R"][b] @ models[ids[e]].sph_pos[g])[2] - models[ids[e]].sph_radius[g]
                     for g, b in enumerate(models[ids[e]].sph_body))
        on_terrain += int(lowest > 0.05)
    assert on_terrain >= n // 2
    print(robot, "terrain: max |state diff| GPU vs oracle over 30 steps: %.2e, %d / %d robots resting on boxes" % (worst, on_terrain, n))


def test_terrain_needs_the_wave_mapping():
    env = _make("MetaAntEnv", [MODELS["ant"]], 4, mapping="lane")
    env.set_terrain(_terrain_for_tests())
    env.reset()
    with pytest.raises(Exception, match="wave mapping"):
        env.step(torch.zeros(4, env.n_joints))


def test_terrain_checkpoint_survives_set_task_and_restores_no_terrain():
    """a checkpoint's terrain boxes become the env's terrain spec — a later set_task() (which rebuilds the params and
    calls _apply_terrain) must neither drop them nor bring back the boxes the env was constructed with; and a checkpoint taken
    WITHOUT terrain removes a terrain the loading env holds."""
    models = [MODELS["ant"]]
    spec = _terrain_for_tests()
    a = _make("MetaAntEnv", models, 4)
    a.set_terrain(spec)
    a.reset(seed=1)
    sd = a.state_dict()
    assert sd["terrain_kind"] == "boxes"
    # (i) an env built with OTHER boxes of the same count, (ii) one built without terrain
    other = [(h, [p[0] + 5.0, p[1], p[2]], q, f) for h, p, q, f in spec]
    for prep in (lambda e: e.set_terrain(other), lambda e: None):
        b = _make("MetaAntEnv", models, 4)
        prep(b)
        b.reset(seed=2)
        b.load_state_dict(sd)
        assert torch.equal(b._terrain_t, a._terrain_t)
        b.set_task(models)                  # rebuilds mg_walker_params: the restored rows must be re-installed
        assert b._terrain_t is not None and torch.equal(b._terrain_t, a._terrain_t)
        assert b._params_c.n_terrain_boxes == len(spec) and b._params_c.terrain == b._terrain_t.data_ptr()
    c = _make("MetaAntEnv", models, 4)
    c.reset(seed=3)
    sd0 = c.state_dict()
    assert sd0["terrain_kind"] == "none"
    a.load_state_dict(sd0)
    assert a._terrain_t is None and a._params_c.n_terrain_boxes == 0
    a.set_task(models)
    assert a._terrain_t is None and a._params_c.n_terrain_boxes == 0


def test_first_reset_rule_is_per_env_and_survives_a_checkpoint():
    """WalkerBaseEnv.reset adds the floor link to robot.parts AFTER the first reset's observation (walker_base_env.py:24-31):
    per env. A masked reset as the first call must leave the other envs' own first reset untouched, and state_dict carries
    which envs have been through one."""
    models = [MODELS["humanoid"], MODELS["humanoid_tra_137"]]
    n = 8
    rs = np.random.RandomState(2)
    n1, n2 = rs.uniform(-0.1, 0.1, (n, 17)), rs.uniform(-0.1, 0.1, (n, 17))
    a, b = _make("MetaHumanoidEnv", models, n), _make("MetaHumanoidEnv", models, n)
    first = a.reset(joint_noise=n1).clone()
    pot_first = a.potential.clone()
    half = torch.arange(n, device="cuda:0") < n // 2
    b.reset(mask=half, joint_noise=n1)
    sd = b.state_dict()                                  # checkpoint between the two partial first resets
    c = _make("MetaHumanoidEnv", models, n)
    c.load_state_dict(sd)
    for env in (b, c):
        ob = env.reset(mask=~half, joint_noise=n1)      # (a masked reset writes the rows it resets: `c` never saw the first half's)
        assert torch.equal(ob[~half], first[~half]) and torch.equal(env.potential, pot_first)       # every env saw ITS first reset
    assert torch.equal(b._obs, first)
    second = a.reset(joint_noise=n2).clone()
    assert not torch.equal(second, a.reset(joint_noise=n1))                           # sanity: noise matters
    a.reset(joint_noise=n2)
    for env in (b, c):
        assert torch.equal(env.reset(joint_noise=n2), second) and torch.equal(env.potential, a.potential)
    fresh = _make("MetaHumanoidEnv", models, n)
    assert not torch.equal(fresh.reset(joint_noise=n2), second)                       # first vs later reset do differ (the floor link)


def test_masked_resets_alone_leave_the_two_launch_path():
    """Envs that are only ever reset through masks: once every env had its first reset, reset() is back to ONE launch — a host
    bitmap follows masks that arrive on the host, the device bitmap is looked at every 64th masked reset otherwise."""
    n = 4
    env = _make("MetaHumanoidEnv", [MODELS["humanoid"]], n)
    rs = np.random.RandomState(0)
    noise = rs.uniform(-0.1, 0.1, (n, 17))
    env.reset(mask=np.array([1, 1, 0, 0], bool), joint_noise=noise)
    assert not env._all_floor_known
    env.reset(mask=[False, False, True, True], joint_noise=noise)
    assert env._all_floor_known
    dev = _make("MetaHumanoidEnv", [MODELS["humanoid"]], n)
    for k in range(64):
        dev.reset(mask=torch.tensor([k % 2 == 0, k % 2 == 1, True, True], device="cuda:0"), joint_noise=noise)
    assert dev._all_floor_known
    assert torch.equal(dev.reset(joint_noise=noise), env.reset(joint_noise=noise))
