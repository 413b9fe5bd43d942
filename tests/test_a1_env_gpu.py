"""GPU: `metagym_amd.quadrupedal.A1GymEnv` — the device-side composition of A1GymEnv.reset / step — against the WHOLE
unmodified reference env recorded on a scripted Bullet client (tests/golden/a1_env.npz): motor commands, 13 x 12 torques per
step, the 37-entry observation, reward terms, reward, done. The physics object replays the recorded world."""
import json
import os

import numpy as np
import pytest
import torch

import metagym_amd
from metagym_amd.quadrupedal import A1GymEnv

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "a1_env.npz")
DEV = "cuda:0"
TOL = dict(rtol=1e-11, atol=1e-11)


class ReplayPhysics(object):
    """Hands the recorded scripted-world states to the env, N identical robots."""

    def __init__(self, g, name, n):
        self.g, self.name, self.n = g, name, n
        self.step_idx, self.sub_idx, self.world_idx = 0, 0, 0
        self.torques = []

    def _t(self, x):
        x = np.asarray(x, np.float64).reshape(-1)
        return torch.as_tensor(np.broadcast_to(x, (self.n, x.size)).copy(), device=DEV)

    def _obs(self, t):
        return self._t(t[0:12]), self._t(t[12:24]), self._t(t[36:40]), self._t(t[40:43])

    def reset(self, mask):
        self.world_idx = 0
        return self._obs(self.g[self.name + "/reset_true_obs"][0])

    def substep(self, torques):
        self.n_sub = getattr(self, "n_sub", 0) + 1
        self.torques.append(torques.cpu().numpy().copy())
        t = self.g[self.name + "/true_obs"][self.step_idx][self.sub_idx]
        self.sub_idx += 1
        if self.sub_idx == 13:
            self.sub_idx, self.step_idx = 0, self.step_idx + 1
            self.world_idx += 1
        return self._obs(t)

    def world(self):
        g, name, k = self.g, self.name, self.world_idx
        return dict(base=self._t(g[name + "/loco_base"][k]), contact=self._t(g[name + "/loco_real_contact"][k]),
                    bad=torch.full((self.n,), int(g[name + "/loco_bad"][k]), dtype=torch.int32, device=DEV),
                    # |normal force| per foot in newtons: GetFootContactsForce('simple') reports it / 100 (a1.py:352-353)
                    foot_force=self._t(g[name + "/loco_contact_force"][k][4:8] * 100.0))


@pytest.mark.parametrize("idx", range(11))
def test_a1_gym_env_matches_the_unmodified_reference_env(idx):
    g = np.load(GOLDEN)
    name, n = str(g["cases"][idx]), 3
    etg, normal, lat_ms, filt = g[name + "/config"]
    spec = json.loads(str(g[name + "/spec"]))      # task terrain, ObservationWrapper entries, yaw target
    d_yaw = spec.get("d_yaw")
    kw = dict(task=spec["task"]) if "task" in spec else {}
    if "sensor_mode" in spec:
        kw["sensor_mode"] = dict({"dis": 1, "motor": 1, "imu": 1, "contact": 1, "footpose": 0, "ETG": 0}, **spec["sensor_mode"])
    phys = ReplayPhysics(g, name, n)
    if spec.get("random_param", {}).get("random_force"):
        # RandomWrapper's pushes: the forces the reference drew from numpy's global stream are inputs (`force_source`), the
        # replayed world records every apply_external_force with the number of sub-steps run so far
        push = g[name + "/push"]
        draws = [(push[i, 6:9], push[i, 3:6]) for i in range(len(push)) if i == 0 or not np.array_equal(push[i, 3:9], push[i - 1, 3:9])]
        calls = {"n": 0}

        def source():
            calls["n"] += 1
            return draws[min(calls["n"] - 1, len(draws) - 1)]       # the first call is reset()'s draw, every later one the redraw at c % 100 == 0
        kw.update(random_param=spec["random_param"], force_source=source)
        phys.pushes, phys.n_sub = [], 0
        phys.apply_external_force = lambda f, p: phys.pushes.append((phys.n_sub, f.cpu().numpy().copy(), p.cpu().numpy().copy()))
    if "dynamic_param" in spec:
        kw["dynamic_param"] = spec["dynamic_param"]
        phys.base_mass = float(g[name + "/reset_dynamics"][0][2])
        phys.set_dynamics = lambda d: None
        lat_ms = -1.0                                               # (the latency comes in through dynamic_param)
    env = metagym_amd.make("quadrupedal-v0", num_envs=n, physics=phys, device=DEV, ETG=int(etg), ETG_w=g[name + "/w"], ETG_b=g[name + "/b"],
                           normal=int(normal), control_latency=0.002 if lat_ms < 0 else 0.001 * lat_ms, filter_=int(filt), **kw)
    assert isinstance(env, A1GymEnv)
    assert env.default_pose[2] == g[name + "/reset_pose_z"][0]                      # 0.28 + the terrain's add_height
    obs, info = env.reset(d_yaw=d_yaw)
    shape = (n,) + g[name + "/reset_obs"][0].shape
    assert np.allclose(obs.cpu().numpy(), np.broadcast_to(g[name + "/reset_obs"][0], shape), **TOL)
    assert np.allclose(np.array(phys.torques)[:, 1], g[name + "/torques"][0], **TOL)
    for k in range(len(g[name + "/action"])):
        assert env.get_time_since_reset() == pytest.approx(g[name + "/t"][k], abs=1e-15)
        phys.torques = []
        a = torch.as_tensor(np.broadcast_to(g[name + "/action"][k], (n, 12)).copy(), device=DEV)
        obs, reward, done, info = env.step(a, d_yaw=d_yaw)
        assert np.allclose(info["real_action"].cpu().numpy()[0], g[name + "/command"][k + 1], **TOL), "%s command, step %d" % (name, k)
        assert np.allclose(np.array(phys.torques)[:, 2], g[name + "/torques"][k + 1], **TOL), "%s torques, step %d" % (name, k)
        assert np.allclose(info["pose"].cpu().numpy()[0], g[name + "/info_pose"][k], **TOL)
        assert np.allclose(info["rot_mat"].cpu().numpy()[0], g[name + "/info_rot_mat"][k], **TOL)
        assert np.allclose(info["footposition"].cpu().numpy()[0], g[name + "/info_footposition"][k].reshape(-1), **TOL)
        assert np.allclose(obs.cpu().numpy(), np.broadcast_to(g[name + "/obs"][k], shape), **TOL), "%s observation, step %d" % (name, k)
        terms = np.array([info[t].cpu().numpy()[1] for t in ("torso", "up", "feet", "tau", "badfoot", "footcontact")])
        assert np.allclose(terms, g[name + "/terms"][k], **TOL), "%s reward terms, step %d" % (name, k)
        assert np.allclose(reward.cpu().numpy(), g[name + "/reward"][k], **TOL)
        assert bool(done.cpu().numpy()[0]) == bool(g[name + "/done"][k])
        if "force_vec" in info and name + "/force_vec" in g.files:
            assert np.allclose(info["force_vec"].cpu().numpy()[0], g[name + "/force_vec"][k], **TOL)
    if name + "/push" in g.files and hasattr(phys, "pushes"):
        # the reference calls applyExternalForce only while a push is on; here every env step hands the physics a (possibly zero)
        # force: the non-zero ones must be the reference's calls — same sub-step count, force and position
        ref = g[name + "/push"]
        mine = [(n_sub, f[0], p_[0]) for n_sub, f, p_ in phys.pushes if np.abs(f).max() > 0]
        # (the reference applies a push right AFTER an env step, for the next one; here it is handed over right BEFORE that next
        # step — so the one after the final recorded step never reaches the physics)
        assert len(mine) == len(ref) - 1
        base = int(ref[0, 0])
        for (n_sub, f, p_), row in zip(mine, ref):
            assert n_sub + base == int(row[0]) and np.allclose(f, row[3:6], **TOL) and np.allclose(p_, row[6:9], **TOL)


EPISODES = os.path.join(os.path.dirname(__file__), "golden", "a1_env_episodes.npz")


class EpisodeReplayPhysics(ReplayPhysics):
    """ReplayPhysics over several episodes of one env object: every reset() hands out the next episode's recorded first
    observation, the world entries run on (reset info, hidden step, steps, next reset info, ...), and the reset pose / heading
    and the terrain the env hands over are recorded for comparison with what the reference handed its simulator."""

    def __init__(self, g, name, n):
        super().__init__(g, name, n)
        self.episode, self.poses, self.terrains = -1, [], []

    def reset(self, mask):
        self.episode += 1
        if self.episode > 0:
            self.world_idx += 1
        return self._obs(self.g[self.name + "/reset_true_obs_all"][self.episode])

    def set_reset_pose(self, pose, mask=None, yaw=None):
        p = torch.as_tensor(pose, dtype=torch.float64).reshape(3, -1).cpu().numpy()
        y = np.zeros(1) if yaw is None else torch.as_tensor(yaw, dtype=torch.float64).reshape(-1).cpu().numpy()
        self.poses.append((p[:, 0].copy(), float(y[0])))

    def set_terrain(self, boxes, default_pose):
        self.terrains.append((len(boxes), list(default_pose)))
        self.set_reset_pose(default_pose)


@pytest.mark.parametrize("idx", range(3))
def test_a1_gym_env_reset_and_step_keywords_match_the_reference(idx):
    """The keyword surface of the reference's reset() / step() over several episodes of ONE env object, against the unmodified
    reference on the scripted client (tests/golden/a1_env_episodes.npz): reset(hardset=True, mode=, stepwidth=, slope=,
    stepheight=, env_vec=) -> a new terrain per episode (reset height, env_info stretches in the reward); reset(yaw=, x_noise=) ->
    the pose and heading handed to the simulator; reset(ETG_w=, ETG_b=); step(donef=); and the reference's own ETG fixture
    (quadrupedal/ESStair_origin.npz as test_ETG.py:5 runs it: task stairstair, ETG=1, zero action, 100 steps)."""
    g = np.load(EPISODES)
    name, n = str(g["cases"][idx]), 3
    etg, normal, lat_ms, filt = g[name + "/config"]
    spec = json.loads(str(g[name + "/spec"]))
    d_yaw = spec.get("d_yaw")
    kw = dict(task=spec["task"]) if "task" in spec else {}
    if "sensor_mode" in spec:
        kw["sensor_mode"] = dict({"dis": 1, "motor": 1, "imu": 1, "contact": 1, "footpose": 0, "ETG": 0}, **spec["sensor_mode"])
    phys = EpisodeReplayPhysics(g, name, n)
    x_draws = iter([p[0] for p, e in zip(g[name + "/reset_pos"], spec["episodes"]) if e["reset_kw"].get("x_noise")])
    # the reference's registered defaults (quadrupedal/__init__.py:9-20) go in as they are
    env = metagym_amd.make("quadrupedal-v0", num_envs=n, physics=phys, device=DEV, action_limit=(0.75, 0.75, 0.75), render=False, on_rack=False,
                           random_dynamic=False, ETG=int(etg), ETG_T=0.5, ETG_H=20, ETG_path="", ETG_w=g[name + "/w"], ETG_b=g[name + "/b"],
                           dynamic_param={}, normal=int(normal), filter_=int(filt), x_noise_source=lambda: next(x_draws), **kw)
    shape = (n,) + g[name + "/reset_obs"][0].shape
    first = list(g[name + "/episode_first_step"]) + [len(g[name + "/action"])]
    sub, new_etg = 0, 0
    for ep_i, ep in enumerate(spec["episodes"]):
        rkw = dict(ep["reset_kw"])
        if ep.get("new_etg"):
            rkw.update(ETG_w=g[name + "/new_etg_w"][new_etg], ETG_b=g[name + "/new_etg_b"][new_etg])
            new_etg += 1
        phys.torques = []
        obs, info = env.reset(d_yaw=d_yaw, **rkw)
        # what reached the simulator: position [add_x, 0, 0.28 + add_height], heading yaw (quaternion z, w = sin, cos of yaw / 2)
        pos, yaw = phys.poses[-1]
        assert np.allclose(pos, g[name + "/reset_pos"][ep_i], rtol=0, atol=1e-15), (ep_i, pos, g[name + "/reset_pos"][ep_i])
        assert np.allclose([np.sin(yaw / 2), np.cos(yaw / 2)], g[name + "/reset_orn"][ep_i][2:], rtol=0, atol=1e-15)
        assert float(info["yaw_init"][0]) == g[name + "/yaw_init"][ep_i]
        rows = g[name + "/env_info"][ep_i][:int(g[name + "/env_info_len"][ep_i])]
        assert len(info["env_info"]) == len(rows)
        for (x0, x1, vec), row in zip(info["env_info"], rows):
            assert np.array_equal(np.asarray([x0, x1] + [float(v) for v in vec]), row)
        assert np.allclose(obs.cpu().numpy(), np.broadcast_to(g[name + "/reset_obs"][ep_i], shape), **TOL), "%s reset observation, episode %d" % (name, ep_i)
        assert np.allclose(np.array(phys.torques)[:, 1], g[name + "/torques"][sub], **TOL)
        sub += 1
        skw = dict(ep.get("step_kw", {}))
        for k in range(first[ep_i], first[ep_i + 1]):
            assert env.get_time_since_reset() == pytest.approx(g[name + "/t"][k], abs=1e-15)
            phys.torques = []
            a = torch.as_tensor(np.broadcast_to(g[name + "/action"][k], (n, 12)).copy(), device=DEV)
            obs, reward, done, info = env.step(a, d_yaw=d_yaw, **skw)
            assert np.allclose(info["real_action"].cpu().numpy()[0], g[name + "/command"][sub], **TOL), "%s command, step %d" % (name, k)
            assert np.allclose(np.array(phys.torques)[:, 2], g[name + "/torques"][sub], **TOL), "%s torques, step %d" % (name, k)
            assert np.allclose(obs.cpu().numpy(), np.broadcast_to(g[name + "/obs"][k], shape), **TOL), "%s observation, step %d" % (name, k)
            terms = np.array([info[t].cpu().numpy()[1] for t in ("torso", "up", "feet", "tau", "badfoot", "footcontact")])
            assert np.allclose(terms, g[name + "/terms"][k], **TOL), "%s reward terms, step %d" % (name, k)
            assert np.allclose(reward.cpu().numpy(), g[name + "/reward"][k], **TOL)
            assert bool(done.cpu().numpy()[0]) == bool(g[name + "/done"][k])
            sub += 1
    assert sub == len(g[name + "/command"])
    if name == "env_hardset_terrains":      # three of the four resets asked for a new terrain; the boxes went to the simulator each time
        assert [t[0] > 0 for t in phys.terrains[1:]] == [True, True, True] and len(phys.terrains) == 4


def test_quadrupedal_constructor_takes_the_reference_signature():
    """a1_gym_env.py:19-40 / quadrupedal/__init__.py:9-20: every keyword of the reference's constructor is accepted; the three
    that select things this package does not build are refused BY NAME, the two that have no effect in the reference either
    (action_limit, step_y) are kept as attributes; action / observation spaces follow simple_openloop.py:124-137."""
    class Null(object):
        def __init__(self, n):
            z = lambda k: torch.zeros(n, k, dtype=torch.float64, device=DEV)
            self.s, self.w = (z(12), z(12), torch.tensor([[0, 0, 0, 1.0]] * n, dtype=torch.float64, device=DEV), z(3)), dict(
                base=z(3), contact=z(4), bad=torch.zeros(n, dtype=torch.int32, device=DEV))
        def reset(self, mask): return self.s
        def substep(self, t): return self.s
        def world(self): return self.w
    env = metagym_amd.make("quadrupedal-v0", num_envs=2, device=DEV, physics=Null(2), action_limit=(0.5, 0.5, 0.5), render=False, on_rack=False,
                           sensor_mode={"dis": 1, "motor": 1, "imu": 1, "contact": 1, "footpose": 0, "ETG": 0}, gait=0, normal=0, filter_=0,
                           action_space=1, random_dynamic=False, ETG=0, ETG_T=0.5, ETG_H=20, ETG_path="", vel_d=0.6, step_y=0.05,
                           task="plane", reward_p=1.0, dynamic_param={}, some_future_keyword=1)
    assert env.action_limit == (0.5, 0.5, 0.5) and env.step_y == 0.05 and env.ignored_kwargs == {"some_future_keyword": 1}
    assert np.allclose(env.action_space.high, [0.1, 0.5, 0.4] * 4) and np.allclose(env.action_space.low, [-0.1, -0.3, -0.6] * 4)
    assert env.observation_space.shape == (37,)
    obs, info = env.reset(yaw=0.0, x_noise=False)
    env.step(torch.zeros(2, 12, dtype=torch.float64, device=DEV), donef=True)
    for bad, pat in ((dict(render=True), "render"), (dict(on_rack=True), "on_rack"), (dict(gait=1), "gait"),
                     (dict(task="heightfield"), "wm_height_out.png")):       # locomotion_gym_env.py:160-164: PyBullet's asset
        with pytest.raises(Exception, match=pat):
            metagym_amd.make("quadrupedal-v0", num_envs=2, device=DEV, physics=Null(2), **bad)
    with pytest.raises(TypeError, match="unexpected keyword"):
        env.reset(yawn=0.3)
    with pytest.raises(Exception, match="set_reset_pose"):
        env.reset(yaw=0.3)                                   # this physics cannot place the robot
    with pytest.raises(Exception, match="terrain_slots"):
        env.configure_reset(torch.tensor([True, False], device=DEV), hardset=True, mode="slope", stepwidth=0.3, slope=0.3, stepheight=0.05, env_vec=[])


def test_unsupported_sensor_modes_are_refused_loudly():
    """What is still refused by name: dynamic_vec without a physics that knows its base mass, values outside
    env_builder.py:62-80; and pushes / simulator-side dynamics on a physics that cannot take them."""
    with pytest.raises(Exception, match="random_force"):
        metagym_amd.make("quadrupedal-v0", num_envs=2, device=DEV, physics=object(), random_param={"random_force": 1})
    with pytest.raises(Exception, match="set_dynamics"):
        metagym_amd.make("quadrupedal-v0", num_envs=2, device=DEV, physics=object(), dynamic_param={"footfriction": 2.0})
    with pytest.raises(Exception, match="random_dynamic"):
        metagym_amd.make("quadrupedal-v0", num_envs=2, device=DEV, physics=object(), random_dynamic=True)
    with pytest.raises(Exception, match="per-link dynamic_param"):    # (needs the per-robot hooks of A1Physics)
        metagym_amd.make("quadrupedal-v0", num_envs=2, device=DEV, physics=object(), dynamic_param={"leginertia": [1.0] * 12})
    with pytest.raises(Exception, match="jointfriction"):             # not one of locomotion_gym_env.py:354-380's keys
        metagym_amd.make("quadrupedal-v0", num_envs=2, device=DEV, physics=object(), dynamic_param={"jointfriction": 0.1})
    for mode in ({"dis": 1, "motor": 1, "imu": 1, "contact": 1, "footpose": 0, "dynamic_vec": 1},
                 {"dis": 1, "motor": 3, "imu": 1, "contact": 1, "footpose": 0}):
        with pytest.raises(Exception, match="sensor_mode"):
            metagym_amd.make("quadrupedal-v0", num_envs=2, device=DEV, physics=object(), sensor_mode=mode)


def test_quadrupedal_without_physics_explains_itself():
    """(neither `physics=` nor `urdf=`: the only case that is refused)"""
    with pytest.raises(Exception, match="a1.urdf"):
        metagym_amd.make("quadrupedal-v0", num_envs=4, device=DEV)


def test_closed_loop_on_the_example_standin_body_holds_the_default_pose():
    """examples/a1_standin: the env's physics protocol driven by this repo's articulated-body engine with a STAND-IN body
    (not the reference's robot). Zero actions -> the PD loop holds (0, 0.9, -1.8) x 4: the base settles near the leg length
    and nothing terminates. A plumbing test of the closed loop, not a parity statement."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "examples", "a1_standin"))
    from physics import StandinPhysics
    n = 256
    env = metagym_amd.make("quadrupedal-v0", num_envs=n, physics=StandinPhysics(n, DEV), device=DEV)
    obs, info = env.reset()
    a = torch.zeros(n, 12, dtype=torch.float64, device=DEV)
    for _ in range(20):
        obs, reward, done, info = env.step(a)
        assert torch.isfinite(obs).all() and not bool(done.any())
    z = info["base"][:, 2]
    assert 0.22 < float(z.min()) and float(z.max()) < 0.30
    assert float((env.robot.GetMotorAngles() - torch.as_tensor([0, 0.9, -1.8] * 4, device=DEV)).abs().max()) < 0.15
    assert float(info["real_contact"].sum(dim=1).min()) >= 2.0      # it stands on its feet (flags flicker with the penetration depth)


@pytest.mark.parametrize("n", [192, 4, 12, 3])
def test_fused_engine_step_equals_thirteen_separate_substeps_bit_for_bit(n):
    """(num_envs = 4, 12, 3 are the sizes at which a `[k, num_envs]` tensor has the shape of a `[num_envs, k]` one — quaternion,
    motor angles, body rate: the layout is declared with `SoA(...)`, never guessed from the shape.)
    The engine's in-launch actuators (13 sub-steps, PD motor model inside the launch, sub-step log -> mg_a1_receive_log)
    against the same closed loop run sub-step by sub-step through mg_a1_apply_action / mg_a1_receive_and_apply (whose motor
    model is pinned to the reference): observations, rewards, torques and the engine state must be bit-identical."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "examples", "a1_standin"))
    from physics import StandinPhysics
    w = np.tile([[0.02], [0.0], [0.015]], (1, 20)) * np.sin(np.linspace(0, 2 * np.pi, 20))
    envs = []
    for fused in (True, False):
        phys = StandinPhysics(n, DEV, fused=fused)
        envs.append(metagym_amd.make("quadrupedal-v0", num_envs=n, physics=phys, device=DEV, ETG=1, ETG_w=w, ETG_b=np.zeros(3),
                                     control_latency=0.0057))
    assert hasattr(envs[0].physics, "fused_step") and not hasattr(envs[1].physics, "fused_step")
    o0, _ = envs[0].reset()
    o1, _ = envs[1].reset()
    assert torch.equal(o0, o1)
    rs = np.random.RandomState(3)
    for k in range(8):
        a = torch.as_tensor(rs.uniform(-0.2, 0.2, (n, 12)), device=DEV)
        r0, r1 = envs[0].step(a), envs[1].step(a)
        assert torch.equal(r0[0], r1[0]), "observation, step %d" % k
        assert torch.equal(r0[1], r1[1]) and torch.equal(r0[2], r1[2])
        assert torch.equal(envs[0].last_torques, envs[1].last_torques), "torques, step %d" % k
        for key in ("pos", "rot", "vel", "omega", "q", "qd"):
            assert torch.equal(getattr(envs[0].physics.env, key), getattr(envs[1].physics.env, key)), key
        assert torch.equal(envs[0].robot.GetControlObservation(), envs[1].robot.GetControlObservation())
    assert torch.isfinite(r0[0]).all()


def test_standin_robot_stands_on_the_task_terrain():
    """`task=` hands the reference's terrain boxes to the physics (StandinPhysics.set_terrain -> mg_walker_params.terrain) and
    resets the robot 0.28 + add_height up: on `slopestair` (start platform 0.278 m above the ground plane) the stand-in holds
    its pose ON the platform; with the boxes withheld it would drop to the plane."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "examples", "a1_standin"))
    from physics import StandinPhysics
    n = 128
    heights = {}
    for with_boxes in (True, False):
        phys = StandinPhysics(n, DEV)
        env = metagym_amd.make("quadrupedal-v0", num_envs=n, physics=phys, device=DEV, task="slopestair")
        assert len(env.terrain_boxes) == 44 and abs(env.add_height - 0.27757873) < 1e-6 and len(env.env_info) == 20
        if not with_boxes:
            phys.env.set_terrain(None)
        obs, info = env.reset()
        a = torch.zeros(n, 12, dtype=torch.float64, device=DEV)
        for _ in range(25):
            obs, reward, done, info = env.step(a)
        assert torch.isfinite(obs).all()
        heights[with_boxes] = float(info["base"][:, 2].mean())
    assert 0.22 + 0.2776 < heights[True] < 0.30 + 0.2776, heights
    assert heights[False] < heights[True] - 0.2, heights


@pytest.mark.parametrize("fused", [True, False])
def test_graph_replay_of_a_step_equals_eager_steps_bit_for_bit(fused):
    """A1GymEnv.capture_step: one env step (ETG, 13 sub-steps on the engine, info, sensors, reward) replayed as a hipGraph
    against the same env stepped eagerly — identical bits in observations, rewards, dones and the engine state, and the
    device-side clock the ETG reads equals the host's get_time_since_reset()."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "examples", "a1_standin"))
    from physics import StandinPhysics
    n = 160
    w = np.tile([[0.02], [0.0], [0.015]], (1, 20)) * np.sin(np.linspace(0, 2 * np.pi, 20))
    rs = np.random.RandomState(4)
    actions = [torch.as_tensor(rs.uniform(-0.1, 0.1, (n, 12)) * (1 + 20 * (np.arange(n) % 4 == 0))[:, None], device=DEV) for _ in range(14)]
    runs = []          # (every fourth robot gets offsets that make it fall: the fused auto-reset runs inside the captured step)
    for graphed in (True, False):
        phys = StandinPhysics(n, DEV, fused=fused)
        env = metagym_amd.make("quadrupedal-v0", num_envs=n, physics=phys, device=DEV, ETG=1, ETG_w=w, ETG_b=np.zeros(3), auto_reset=True)
        env.reset()
        step = env.step
        if graphed:
            step = env.capture_step()
            env.reset()
        rec = []
        for a in actions:
            obs, reward, done, info = step(a)
            assert float(env._substeps_dev[1]) * 0.002 == env.get_time_since_reset()      # (robot 1 is never reset)
            rec.append((obs.clone(), reward.clone(), done.clone(), phys.env.q.clone(), phys.env.pos.clone()))
        runs.append(rec)
    assert any(bool(r[2].any()) for r in runs[0]), "no episode ended: the auto-reset path was not exercised"
    for k, (g, e) in enumerate(zip(*runs)):
        for x, y in zip(g, e):
            assert torch.equal(x, y), "step %d" % k


@pytest.mark.parametrize("fused", [True, False])
def test_partial_reset_inside_a_step_equals_a_fresh_reset_for_those_robots(fused):
    """step(reset_mask=m): the robots in m run A1GymEnv.reset() inside the step (its hidden zero-action env step IS this step),
    the others must not notice. Against two witnesses on the same deterministic physics: an env that is never reset (the
    untouched robots, bit for bit) and a fresh env that does a full reset() at that moment (the reset robots: the observation
    reset() returns, then every later observation / reward / done, bit for bit)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "examples", "a1_standin"))
    from physics import StandinPhysics
    n, k_reset, k_end = 96, 5, 10
    w = np.tile([[0.02], [0.0], [0.015]], (1, 20)) * np.sin(np.linspace(0, 2 * np.pi, 20))
    rs = np.random.RandomState(9)
    actions = [torch.as_tensor(rs.uniform(-0.1, 0.1, (n, 12)), device=DEV) for _ in range(k_end)]
    m = torch.zeros(n, dtype=torch.bool, device=DEV)
    m[::3] = True

    def make():
        env = metagym_amd.make("quadrupedal-v0", num_envs=n, physics=StandinPhysics(n, DEV, fused=fused), device=DEV, ETG=1, ETG_w=w,
                               ETG_b=np.zeros(3), sensor_mode={"dis": 1, "motor": 1, "imu": 1, "contact": 1, "footpose": 0, "ETG": 1,
                                                               "RNN": {"time_steps": 2, "time_interval": 2, "mode": "stack"}})
        return env

    a_env, never, fresh = make(), make(), make()
    a_env.reset(); never.reset()
    for k in range(k_end):
        if k == k_reset:
            out = a_env.step(actions[k], reset_mask=m)
            obs_f, _ = fresh.reset()
            assert torch.equal(out[0][m], obs_f[m]), "the observation reset() returns"
            assert float(out[1][m].abs().max()) == 0.0 and not bool(out[2][m].any()) and bool(out[3]["reset"][m].all())
        else:
            out = a_env.step(actions[k])
            if k > k_reset:
                of, rf, df, _ = fresh.step(actions[k])
                assert torch.equal(out[0][m], of[m]) and torch.equal(out[1][m], rf[m]) and torch.equal(out[2][m], df[m]), "reset robots, step %d" % k
        on, rn, dn, _ = never.step(actions[k])
        assert torch.equal(out[0][~m], on[~m]) and torch.equal(out[1][~m], rn[~m]) and torch.equal(out[2][~m], dn[~m]), "untouched robots, step %d" % k


def test_auto_reset_restarts_finished_robots_without_a_host_round_trip():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "examples", "a1_standin"))
    from physics import StandinPhysics
    n = 64
    env = metagym_amd.make("quadrupedal-v0", num_envs=n, physics=StandinPhysics(n, DEV), device=DEV, auto_reset=True)
    env.reset()
    rs = np.random.RandomState(2)
    wild = torch.as_tensor(rs.uniform(-2.5, 2.5, (n, 12)), device=DEV)        # large offsets: robots fall over and terminate
    calm = torch.zeros(n, 12, dtype=torch.float64, device=DEV)
    ended = torch.zeros(n, dtype=torch.bool, device=DEV)
    prev_done = torch.zeros(n, dtype=torch.bool, device=DEV)
    for k in range(60):
        a = torch.where((torch.arange(n, device=DEV) % 2 == 0).reshape(-1, 1), wild, calm)
        obs, reward, done, info = env.step(a)
        assert torch.isfinite(obs).all()
        assert torch.equal(info["reset"], prev_done)                           # exactly the robots that ended one step earlier
        ended |= done
        prev_done = done.clone()
    assert bool(ended[0::2].any()) and not bool(ended[1::2].any())
