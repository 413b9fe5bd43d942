#!/usr/bin/env python3
"""Golden vectors for the Quadrupedal (A1) SENSOR STACK, recorded from the UNMODIFIED reference.

TEST INFRASTRUCTURE; runs only in the build container (needs /root/reference):

    python oracle/gen_golden_a1_sensors.py

The observation `A1GymEnv.step` returns is assembled in Python from the robot's getters by the sensors
`env_builder.build_regular_env` attaches (envs/env_builder.py:62-80, SENSOR_MODE dis / imu / motor / contact = 1):
    BaseDisplacementSensor(convert_to_local_frame=True)   envs/sensors/robot_sensors.py:217-312
    IMUSensor(channels R P Y dR dP dY)                     :314-437
    MotorAngleAccSensor(num_motors=12, dt)                 :85-162
    FootContactSensor                                      :552-578
ordered by sensor name (`LocomotionGymEnv._get_observation`, locomotion_gym_env.py:621-632) and flattened
(`env_utils.flatten_observations`, envs/utilities/env_utils.py:11-42). Here the real sensor objects and those two
functions run on a scripted robot (every getter value is recorded as an INPUT), with the reference's call order:
reset(): sensor.reset() ... on_reset ... get_observation; step(): on_step ... get_observation
(locomotion_gym_env.py:231-232,426-427,521-522,546)."""
import collections
import collections.abc
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("METAGYM_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden", "a1_sensors.npz")
OUT_NOISE = os.path.join(ROOT, "tests", "golden", "a1_sensors_noise.npz")
# sensor_mode["noise"] (env_builder.py:60-71): every sensor adds np.random.normal draws inside _get_observation. The noise cases
# record them — np.random.normal is wrapped while the reference's sensors run — as 33 values per observation, slot order
# displacement 3 (sigma 1e-2), rpy 3 (6e-2), drpy 3 (1e-1), motor angles 12 (1e-2), motor rates 12 (0.5)
# (robot_sensors.py:281-284, 399-402, 146-148), so a replay needs no knowledge of numpy's stream.
NOISE_SLOTS = {(1e-2, None): (0, 1), (6e-2, 3): (3, 3), (1e-1, 3): (6, 3), (1e-2, 12): (9, 12), (0.5, 12): (21, 12)}


class Robot(object):
    def GetBasePosition(self): return self.base
    def GetBaseRollPitchYaw(self): return self.rpy.copy()
    def GetBaseRollPitchYawRate(self): return self.drpy.copy()
    def GetMotorAngles(self): return self.angles.copy()
    def GetFootContactsForce(self, mode="simple"): return np.concatenate([self.contact, self.force])


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference not mounted at %s — run in the build container" % REF)
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(HERE, "refstubs")); sys.path.insert(0, REF)
    np.int = int
    collections.Sequence = collections.abc.Sequence
    import metagym.quadrupedal  # noqa: F401
    from metagym.quadrupedal.envs import locomotion_gym_env
    from metagym.quadrupedal.envs.sensors import robot_sensors
    from metagym.quadrupedal.envs.utilities import env_utils
    record([dict(name="sensors_raw", normal=0, seed=1), dict(name="sensors_normalised", normal=1, seed=2)], OUT, False,
           locomotion_gym_env, robot_sensors, env_utils)
    record([dict(name="sensors_noise_raw", normal=0, seed=3), dict(name="sensors_noise_normalised", normal=1, seed=4)], OUT_NOISE, True,
           locomotion_gym_env, robot_sensors, env_utils)


def record(cases, out_path, noise, locomotion_gym_env, robot_sensors, env_utils):
    out = {"numpy_version": np.array(np.__version__)}
    for c in cases:
        rs = np.random.RandomState(c["seed"])
        dt = 13 * 0.002
        sensors = [robot_sensors.BaseDisplacementSensor(convert_to_local_frame=True, normal=c["normal"], noise=noise),
                   robot_sensors.IMUSensor(channels=["R", "P", "Y", "dR", "dP", "dY"], normal=c["normal"], noise=noise),
                   robot_sensors.MotorAngleAccSensor(num_motors=12, normal=c["normal"], noise=noise, dt=dt),
                   robot_sensors.FootContactSensor()]
        robot = Robot()
        for s in sensors:
            s.set_robot(robot)

        class Env(object):
            def all_sensors(self): return sensors
        env = Env()
        rec = collections.defaultdict(list)
        drawn = np.zeros(33)
        real_normal = np.random.normal

        def recording_normal(loc, scale, size=None):
            v = real_normal(loc, scale, size)
            start, count = NOISE_SLOTS[(scale, size)]
            if size is None:                                     # the three scalar draws of BaseDisplacementSensor, in dx dy dz order
                start += recording_normal.scalars
                recording_normal.scalars += 1
            drawn[start:start + count] = v
            return v

        def observe():
            drawn[:] = 0.0
            recording_normal.scalars = 0
            np.random.normal = recording_normal
            try:
                obs = env_utils.flatten_observations(locomotion_gym_env.LocomotionGymEnv._get_observation(env))[0]
            finally:
                np.random.normal = real_normal
            if noise:
                rec["in_noise"].append(drawn.copy())
            return obs
        np.random.seed(c["seed"] + 100)

        def world(k):
            robot.base = tuple(np.array([0.02 * k, 0.003 * k, 0.27]) + rs.uniform(-0.01, 0.01, 3))
            robot.rpy = np.array([rs.uniform(-0.2, 0.2), rs.uniform(-0.2, 0.2), 0.3 + 0.02 * k + rs.uniform(-0.01, 0.01)])
            robot.drpy = rs.uniform(-1, 1, 3)
            robot.angles = np.array([0, 0.9, -1.8] * 4) + rs.uniform(-0.3, 0.3, 12)
            robot.contact = (rs.rand(4) < 0.6).astype(np.float64)
            robot.force = rs.uniform(0, 1, 4)
            for key in ("base", "rpy", "drpy", "angles", "contact"):
                rec["in_" + key].append(np.array(getattr(robot, key), dtype=np.float64))

        for episode in range(2):
            world(0)
            for s in sensors:
                s.reset()                                        # locomotion_gym_env.py:231-232
            for s in sensors:
                s.on_reset(env)                                  # :426-427
            obs = observe()
            rec["kind"].append(0)
            rec["obs"].append(obs)
            for k in range(1, 15):
                world(k)
                for s in sensors:
                    s.on_step(env)                               # :521-522
                obs = observe()
                rec["kind"].append(1)
                rec["obs"].append(obs)
        for k, v in rec.items():
            out[c["name"] + "/" + k] = np.array(v)
        out[c["name"] + "/config"] = np.array([c["normal"], dt], dtype=np.float64)
    out["cases"] = np.array([c["name"] for c in cases])
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, os.path.getsize(out_path), "bytes; obs", out[cases[0]["name"] + "/obs"].shape)


if __name__ == "__main__":
    main()
