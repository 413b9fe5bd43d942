#!/usr/bin/env python3
"""Golden vectors for the WHOLE `A1GymEnv.reset / step` composition, recorded from the UNMODIFIED reference.

TEST INFRASTRUCTURE; runs only in the build container (needs /root/reference):

    python oracle/gen_golden_a1_env.py

`metagym.quadrupedal.A1GymEnv` (envs/gym_envs/a1_gym_env.py) is constructed exactly as `gym.make('quadrupedal-v0', ...)`
would (env_builder.build_regular_env -> LocomotionGymEnv -> a1.A1 with the four default sensors -> obs-to-array ->
trajectory generator -> ETGWrapper -> ActionFilterWrapper -> RandomWrapper -> ObservationWrapper -> RewardShaping,
MonitorEnv.py:14-25) — with one substitution: its `BulletClient` is the scripted 12-joint world of gen_golden_a1.py,
extended by the world-level calls the env makes (plane, gravity, contacts, debug items). PyBullet and a1.urdf are not in
the reference tree; the scripted world decides nothing the checkers are graded on: per sub-step and per env step every
value it hands out (joint states, base pose / rates, contact points) is recorded as an INPUT.

What this pins beyond the per-piece goldens (a1_actuation / a1_control / a1_sensors / a1_filter): the ORDER in which
`A1GymEnv.step` composes them — which time the ETG sees, that the policy action is added to the PREVIOUS ETG output,
13 x (ApplyAction, stepSimulation, ReceiveObservation), sensors' on_step before the observation, the reward computed from
this step's info against last step's base / feet, reset()'s hidden first step with a zero action (MonitorEnv.py:310).
Recorded per step: policy action, motor command reaching robot.Step, the 13 x 12 torques, observation (37), reward,
done, the reward terms, and the world inputs."""
import collections
import collections.abc
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import gen_golden_a1 as ga  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "a1_env.npz")


class ScriptedWorld(ga.ScriptedBullet):
    """gen_golden_a1.ScriptedBullet + the world-level API of LocomotionGymEnv / MonitorEnv, with a1.urdf's joint list
    (imu, then hip / hip_fixed / upper / lower / toe per leg: toe links 5, 10, 15, 20 as GetBadFootContacts assumes,
    robots/a1.py:330)."""
    JOINTS = ["imu_joint"]
    for leg in ("FR", "FL", "RR", "RL"):
        JOINTS += ["%s_hip_joint" % leg, "%s_hip_fixed" % leg, "%s_upper_joint" % leg, "%s_lower_joint" % leg, "%s_toe_fixed" % leg]
    COV_ENABLE_RENDERING, COV_ENABLE_GUI, COV_ENABLE_SINGLE_STEP_RENDERING, DIRECT, GUI = 7, 1, 13, 2, 1
    LINK_FRAME = 1

    def __init__(self, seed, dt=0.002):
        super(ScriptedWorld, self).__init__(seed, dt)
        self.records = collections.defaultdict(list)
        self.v_base = np.array([0.4, 0.02, 0.0])
        self.contacts = []

    def setAdditionalSearchPath(self, p): pass
    def resetSimulation(self): pass
    def setPhysicsEngineParameter(self, **k): pass
    def setTimeStep(self, t): self.dt = t
    def setGravity(self, *g): pass
    def configureDebugVisualizer(self, *a, **k): pass
    def resetDebugVisualizerCamera(self, *a, **k): pass
    def addUserDebugLine(self, **k): return 0
    def removeUserDebugItem(self, i): pass
    def removeAllUserDebugItems(self): pass
    def loadURDF(self, path, *a, **k): return 0 if "plane" in str(path) else 1
    def resetBasePositionAndOrientation(self, body, pos, orn):
        self.base_pos = list(pos)
        self.base_pos_at_reset = list(pos)
        self.base_orn_at_reset = list(orn)
    def getMatrixFromQuaternion(self, q):
        x, y, z, w = q
        return (1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z),
                2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y))
    def getContactPoints(self, bodyA=None):
        return list(self.contacts)
    def applyExternalForce(self, objectUniqueId=None, linkIndex=None, forceObj=None, posObj=None, flags=None):
        # RandomWrapper's pushes (MonitorEnv.py:530-535,644-660): recorded with the number of stepSimulation calls so far, so the
        # checkers know which sub-step each one precedes; the scripted world itself does not react to them
        self.records["push"].append(np.concatenate(([self.n_sim_steps, linkIndex, flags], np.asarray(forceObj, float), np.asarray(posObj, float))))
    n_sim_steps = 0

    def stepSimulation(self):
        self.n_sim_steps += 1
        super(ScriptedWorld, self).stepSimulation()
        self.base_pos = list(np.asarray(self.base_pos) + self.dt * (self.v_base + 0.2 * np.sin(3 * self.t + np.arange(3))))
    def new_contacts(self):
        """A fresh scripted contact set for this env step: some feet down, now and then a knee or the trunk."""
        pts = []
        for link in (5, 10, 15, 20):
            if self.rs.rand() < 0.75:
                pts.append((0, 1, 0, link, -1, (0, 0, 0), (0, 0, 0), (0.0, 0.0, 1.0), 0.0, float(self.rs.uniform(5, 60)), 0, (1, 0, 0), 0, (0, 1, 0)))
        if self.rs.rand() < 0.2:
            pts.append((0, 1, 0, int(self.rs.choice([4, 9, -1, 14])), -1, (0, 0, 0), (0, 0, 0), (0.0, 0.0, 1.0), 0.0, 10.0, 0, (1, 0, 0), 0, (0, 1, 0)))
        if self.rs.rand() < 0.1:
            pts.append((0, 1, 1, 5, 3, (0, 0, 0), (0, 0, 0), (0.0, 0.0, 1.0), 0.0, 1.0, 0, (1, 0, 0), 0, (0, 1, 0)))   # a self contact: ignored
        self.contacts = pts


CASES = [dict(name="env_etg_traj", seed=1, ETG=1, wscale=0.04, n_steps=25, normal=0),
         dict(name="env_plain", seed=2, ETG=0, wscale=0.0, n_steps=20, normal=1),
         dict(name="env_latency", seed=3, ETG=1, wscale=0.05, n_steps=20, normal=0, dynamic_param={"control_latency": 13.7}),
         dict(name="env_action_filter", seed=4, ETG=1, wscale=0.04, n_steps=20, normal=0, filter_=1),
         # task terrains: env_info comes from the reference's own terrain builder (its pybullet calls go to a recorder)
         # and the scripted base runs fast enough to cross flat / up-slope / flat / down-slope stretches
         dict(name="env_slopeslope", seed=5, ETG=1, wscale=0.04, n_steps=40, normal=0, task="slopeslope", v_base=[2.6, 0.02, 0.0]),
         dict(name="env_slopestair", seed=6, ETG=0, wscale=0.0, n_steps=40, normal=0, task="slopestair", v_base=[3.4, -0.01, 0.0]),
         # ObservationWrapper's own entries (MonitorEnv.py:136-221): ETG output (normalised), ETG_obs, yaw target, RNN frames
         dict(name="env_obs_extras_stack", seed=7, ETG=1, wscale=0.04, n_steps=16, normal=1, d_yaw=0.3,
              sensor_mode={"ETG": 1, "ETG_obs": 1, "yaw": 1, "RNN": {"time_steps": 2, "time_interval": 3, "mode": "stack"}}),
         dict(name="env_obs_extras_gru", seed=8, ETG=1, wscale=0.04, n_steps=10, normal=0, d_yaw=-0.2,
              sensor_mode={"ETG": 1, "yaw": 1, "RNN": {"time_steps": 3, "time_interval": 1, "mode": "GRU"}}),
         # the other sensors env_builder.py:62-80 can pick: MotorAngleSensor (motor 2), the rate-only IMU (imu 2), SimpleFootForceSensor
         # (contact 2), FootPoseSensor (normalised), and sensors switched off
         dict(name="env_sensors_alt", seed=9, ETG=1, wscale=0.04, n_steps=14, normal=1,
              sensor_mode={"dis": 0, "motor": 2, "imu": 2, "contact": 2, "footpose": 1}),
         dict(name="env_sensors_min", seed=10, ETG=0, wscale=0.0, n_steps=12, normal=0,
              sensor_mode={"dis": 1, "motor": 0, "imu": 0, "contact": 0, "footpose": 1, "ETG_obs": 1}),
         # RandomWrapper's pushes (a new force every 100 env steps, applied for 50) and explicit dynamics (`dynamic_param`:
         # control latency in ms, foot friction, base-mass ratio), with the observation entries that report them
         dict(name="env_random_force", seed=11, ETG=1, wscale=0.03, n_steps=108, normal=0, random_param={"random_dynamics": 0, "random_force": 1},
              dynamic_param={"control_latency": 17.0, "footfriction": 1.7, "basemass": 1.1},
              sensor_mode={"ETG": 1, "force_vec": 1, "dynamic_vec": 1, "yaw": 1}, d_yaw=0.1)]

# Round 4: the keyword surface of reset() / step() (locomotion_gym_env.py:297-338, MonitorEnv.py:246-260,343) over SEVERAL
# episodes of one env object, written to a1_env_episodes.npz (a1_env.npz above stays byte-identical):
_UPSTAIR, _DOWNSLOPE, _PLANE = [0, 0, 1, 0, 0, 0.08, 0.25], [0, 1, 0, 0, 0.34, 0, 0], [0, 0, 0, 0, 0, 0, 0]
EPISODE_CASES = [
    # a new terrain per episode: reset(hardset=True, mode=, stepwidth=, slope=, stepheight=, env_vec=) rebuilds add_height / env_info
    dict(name="env_hardset_terrains", seed=21, ETG=1, wscale=0.04, normal=0, v_base=[2.4, 0.01, 0.0], episodes=[
        dict(n_steps=28, reset_kw=dict(hardset=True, mode="special", stepwidth=0.3, slope=0.34, stepheight=0.07,
                                       env_vec=[_UPSTAIR, _DOWNSLOPE, _PLANE] * 3)),
        dict(n_steps=24, reset_kw=dict(hardset=True, mode="downstair", stepwidth=0.28, slope=0.3, stepheight=0.06, env_vec=[])),
        dict(n_steps=12, reset_kw=dict(hardset=False, mode="slope", stepwidth=0.3, slope=0.4, stepheight=0.05, env_vec=[])),   # hardset False: the terrain stays
        dict(n_steps=14, reset_kw=dict(hardset=True, mode="slope", stepwidth=0.3, slope=0.4, stepheight=0.05, env_vec=[]))]),
    # start heading and position noise: reset(yaw=, x_noise=); new ETG weights for the second episode: reset(ETG_w=, ETG_b=); step(donef=)
    dict(name="env_yaw_xnoise_etg", seed=22, ETG=1, wscale=0.04, normal=0, d_yaw=0.25, sensor_mode={"yaw": 1}, episodes=[
        dict(n_steps=14, reset_kw=dict(yaw=0.4, x_noise=True), step_kw=dict(donef=True)),
        dict(n_steps=14, reset_kw=dict(yaw=-0.7, x_noise=True), new_etg=0.06),
        dict(n_steps=8, reset_kw=dict(x_noise=False))]),
    # the reference's own ETG fixture exactly as quadrupedal/test_ETG.py:5 uses it: task="stairstair", ETG=1,
    # ETG_path="ESStair_origin.npz", zero policy action, 100 steps
    dict(name="env_etg_fixture", seed=23, ETG=1, etg_file="ESStair_origin.npz", normal=0, task="stairstair", v_base=[0.5, 0.0, 0.0],
         zero_action=True, episodes=[dict(n_steps=100, reset_kw={})])]


def main():
    record(CASES, OUT)
    record(EPISODE_CASES, OUT.replace("a1_env.npz", "a1_env_episodes.npz"), episodes=True)


def record(cases, out_path, episodes=False):
    a1, robot_config = ga.import_reference()
    import pybullet_utils.bullet_client as bullet_client
    from metagym.quadrupedal.envs.gym_envs import a1_gym_env
    from metagym.quadrupedal.robots import minitaur
    out = {"numpy_version": np.array(np.__version__)}
    from metagym.quadrupedal.envs.utilities import terrain
    from gen_golden_a1_terrain import Recorder
    for c in cases:
        world = ScriptedWorld(c["seed"])
        if "v_base" in c:
            world.v_base = np.array(c["v_base"], dtype=np.float64)
        terrain.p = Recorder()
        bullet_client.BulletClient = lambda connection_mode=None, w=world: w
        rs = np.random.RandomState(100 + c["seed"])
        H = 20
        if "etg_file" in c:       # a file the reference ships (quadrupedal/ESStair_origin.npz), loaded by ETGWrapper itself
            path = os.path.join(os.path.dirname(a1_gym_env.__file__), "..", "..", c["etg_file"])
            saved = np.load(path)
            w, b = saved["w"], saved["b"]
        else:
            w, b = rs.uniform(-1, 1, (3, H)) * c["wscale"], rs.uniform(-1, 1, 3) * c["wscale"] * 0.2
            path = "/tmp/_etg_env_%s.npz" % c["name"]
            np.savez(path, w=w, b=b)
        rec = collections.defaultdict(list)
        # record what robot.Step receives and what each sub-step sees / produces, through the robot class's own methods
        orig_step, orig_apply, orig_recv = minitaur.Minitaur.Step, minitaur.Minitaur.ApplyAction, minitaur.Minitaur.ReceiveObservation
        cur = {"torques": [], "true": []}

        def step_rec(self, action, control_mode=None):
            cur["torques"], cur["true"] = [], []
            world.new_contacts()
            rec["command"].append(np.array(action, dtype=np.float64))
            r = orig_step(self, action, control_mode)
            rec["torques"].append(np.array(cur["torques"]))
            rec["true_obs"].append(np.array(cur["true"]))
            return r

        def apply_rec(self, cmd, mode):
            t = orig_apply(self, cmd, mode)
            cur["torques"].append(np.array(t, dtype=np.float64))
            return t

        def recv_rec(self):
            orig_recv(self)
            cur["true"].append(np.array(self._observation_history[0], dtype=np.float64))
            rec["all_true_obs"].append(np.array(self._observation_history[0], dtype=np.float64))
        minitaur.Minitaur.Step, minitaur.Minitaur.ApplyAction, minitaur.Minitaur.ReceiveObservation = step_rec, apply_rec, recv_rec
        # every info dictionary LocomotionGymEnv hands up after construction, in call order: reset, the hidden zero-action
        # step of RewardShaping.reset (MonitorEnv.py:310), then one per env.step
        from metagym.quadrupedal.envs import locomotion_gym_env
        orig_lreset, orig_lstep = locomotion_gym_env.LocomotionGymEnv.reset, locomotion_gym_env.LocomotionGymEnv.step
        live = {"on": False}

        def note(info, kind, robot):
            if not live["on"]:
                return
            rec["loco_kind"].append(kind)
            for key in ("base", "pose", "rot_mat", "footposition", "real_contact", "energy", "drpy", "joint_angle"):
                rec["loco_" + key].append(np.array(info[key], dtype=np.float64))
            rec["loco_bad"].append(robot.GetBadFootContacts())
            rec["loco_contact_force"].append(np.array(robot.GetFootContactsForce(mode="simple"), dtype=np.float64))

        def lreset(self, **kw):
            r = orig_lreset(self, **kw)
            note(r[1], 0, self._robot)
            return r

        def lstep(self, action):
            r = orig_lstep(self, action)
            note(r[3], 1, self._robot)
            return r
        locomotion_gym_env.LocomotionGymEnv.reset, locomotion_gym_env.LocomotionGymEnv.step = lreset, lstep
        try:
            np.random.seed(1000 + c["seed"])      # RandomWrapper draws its pushes from numpy's global stream; they are recorded as inputs
            env = a1_gym_env.A1GymEnv(ETG=c["ETG"], ETG_path=path, normal=c["normal"], dynamic_param=c.get("dynamic_param", {}),
                                      filter_=c.get("filter_", 0), task=c.get("task", "plane"),
                                      **({"random_param": c["random_param"]} if "random_param" in c else {}),
                                      sensor_mode=dict({"dis": 1, "motor": 1, "imu": 1, "contact": 1, "footpose": 0, "ETG": 0},
                                                       **c.get("sensor_mode", {})))
            step_kw = {"d_yaw": c["d_yaw"]} if "d_yaw" in c else {}
            for ep_i, ep in enumerate(c["episodes"] if episodes else [dict(n_steps=c["n_steps"], reset_kw={})]):
                n_before_reset = len(rec["all_true_obs"])
                live["on"] = True
                reset_kw = dict(step_kw, **ep["reset_kw"])
                if ep.get("new_etg"):                                      # ETGWrapper.reset(ETG_w=, ETG_b=) MonitorEnv.py:250-253
                    reset_kw["ETG_w"] = rs.uniform(-1, 1, (3, H)) * ep["new_etg"]
                    reset_kw["ETG_b"] = rs.uniform(-1, 1, 3) * ep["new_etg"] * 0.2
                    rec["new_etg_w"].append(reset_kw["ETG_w"])
                    rec["new_etg_b"].append(reset_kw["ETG_b"])
                obs, info = env.reset(**reset_kw)
                rec["reset_pose_z"].append(world.base_pos_at_reset[2])
                if episodes:
                    rec["reset_pos"].append(np.array(world.base_pos_at_reset, dtype=np.float64))
                    rec["reset_orn"].append(np.array(world.base_orn_at_reset, dtype=np.float64))
                    # robot.Reset's one observation: the one right before the hidden step's 13. (With hardset=True the reference
                    # first rebuilds its whole world and robot, locomotion_gym_env.py:243-276 — resetSimulation(), a new robot object
                    # whose constructor settles for 2 s — so 500-odd observations of the discarded settle phase come before it.)
                    rec["reset_true_obs_all"].append(rec["all_true_obs"][len(rec["all_true_obs"]) - 14])
                    rec["episode_first_step"].append(len(rec["obs"]))
                    ei = info["env_info"]
                    rec["env_info_len"].append(len(ei))
                    rows = np.zeros((32, 9))
                    for i, (x0, x1, vec) in enumerate(ei):
                        rows[i] = [x0, x1] + [float(v) for v in vec]
                    rec["env_info"].append(rows)
                    rec["yaw_init"].append(float(info["yaw_init"]))
                rec["reset_force_vec"].append(np.array(info.get("force_vec", np.zeros(6)), dtype=np.float64))
                rec["reset_dynamics"].append(np.array(info.get("dynamics", np.zeros(3)), dtype=np.float64))
                rec["n_sim_steps_after_reset"].append(world.n_sim_steps)
                rec["reset_obs"].append(np.array(obs, dtype=np.float64))
                rec["n_true_obs_before_reset"].append(n_before_reset)
                rec["n_true_obs_after_reset"].append(len(rec["all_true_obs"]))
                rec["n_commands_after_reset"].append(len(rec["command"]))
                ep_step_kw = dict(step_kw, **ep.get("step_kw", {}))
                for k in range(ep["n_steps"]):
                    action = np.zeros(12) if c.get("zero_action") else rs.uniform(-0.3, 0.3, 12)
                    rec["action"].append(action)
                    rec["t"].append(env.get_time_since_reset())
                    obs, reward, done, info = env.step(action, **ep_step_kw)
                    rec["obs"].append(np.array(obs, dtype=np.float64))
                    rec["reward"].append(float(reward))
                    rec["done"].append(bool(done))
                    rec["terms"].append(np.array([info[t] for t in ("torso", "up", "feet", "tau", "badfoot", "footcontact")]))
                    for key in ("base", "pose", "rot_mat", "footposition", "real_contact", "energy", "drpy", "joint_angle"):
                        rec["info_" + key].append(np.array(info[key], dtype=np.float64))
                    rec["bad"].append(env.robot.GetBadFootContacts())
                    rec["force_vec"].append(np.array(info.get("force_vec", np.zeros(6)), dtype=np.float64))
        finally:
            locomotion_gym_env.LocomotionGymEnv.reset, locomotion_gym_env.LocomotionGymEnv.step = orig_lreset, orig_lstep
            minitaur.Minitaur.Step, minitaur.Minitaur.ApplyAction, minitaur.Minitaur.ReceiveObservation = orig_step, orig_apply, orig_recv
        if world.records["push"]:
            rec["push"] = world.records["push"]
        rec["reset_true_obs"] = [rec["all_true_obs"][rec["n_true_obs_before_reset"][0]]]      # robot.Reset's one observation
        del rec["all_true_obs"]                                                               # (500 settle sub-steps of __init__)
        for k2, v in rec.items():
            out[c["name"] + "/" + k2] = np.array(v)
        out[c["name"] + "/w"], out[c["name"] + "/b"] = w, b
        out[c["name"] + "/spec"] = np.array(json.dumps({k: c[k] for k in ("task", "sensor_mode", "d_yaw", "random_param", "dynamic_param", "episodes",
                                                                            "zero_action", "etg_file") if k in c}))
        out[c["name"] + "/config"] = np.array([c["ETG"], c["normal"], c.get("dynamic_param", {}).get("control_latency", -1.0),
                                               c.get("filter_", 0)], dtype=np.float64)
        print(c["name"], "steps", len(rec["obs"]), "obs dim", np.array(rec["obs"]).shape, "dones", int(np.sum(rec["done"])),
              "reward range", np.min(rec["reward"]), np.max(rec["reward"]))
    out["cases"] = np.array([c["name"] for c in cases])
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, os.path.getsize(out_path), "bytes")


if __name__ == "__main__":
    main()
