"""The terrains behind `quadrupedal-v0`'s `task=` (metagym/quadrupedal/envs/utilities/terrain.py and the task table in
envs/locomotion_gym_env.py:25-41,309-325), as data.

The reference builds each terrain by creating static boxes in its PyBullet world and, alongside, the two things the
Python side of the env uses: `add_height` (the robot is reset to z = 0.28 + add_height, locomotion_gym_env.py:337) and
`env_info`, the list of `[x_start, x_end, env_vec[7]]` stretches RewardShaping looks up by the base's x
(MonitorEnv.py:324-327, :416-419, :451-454, :500-503). There is no PyBullet here, so `upstair_terrain` returns the boxes
instead of creating them:

    add_height, env_info, boxes = upstair_terrain(mode="special", env_vecs=TASK_ENV_VECS["stairstair"])
    boxes: list of Box(half_extents[3], position[3], quaternion[4] (x, y, z, w), friction)

in creation order — what a batched physics needs to build the same ground. `env_info` feeds `RewardShaping(env_info=)`
unchanged. Pinned by tests/golden/a1_terrain.npz: the unmodified reference module run on a recording `pybullet`
(oracle/gen_golden_a1_terrain.py); boxes, add_height and env_info are bit-identical for every task and mode.

Kept quirks: `balance_beam` sets the friction of its first box twice and never that of the second (terrain.py:254-258),
so the second keeps the engine default; "stair-var" uses friction 1.0 and reports no env_info; in "special"/random
modes the env vectors are normalised in place (a slope's step fields and a stair's slope field are cleared)."""
import collections
import math

import numpy as np

STEP_HEIGHT = np.arange(0.08, 0.101, 0.002)          # terrain.py:6-11
SLOPE = np.arange(0.3, 0.501, 0.02)
STEP_WIDTH = np.arange(0.26, 0.401, 0.02)
STEP_PER_NUM = 5
DELTA_X = 1
FRICTION = 5.0
DEFAULT_FRICTION = 0.5      # Bullet's lateral friction for a body nobody called changeDynamics on

Box = collections.namedtuple("Box", "half_extents position quaternion friction")

_upstair = (0, 0, 1, 0, 0, 0.08, 0.25)               # locomotion_gym_env.py:29-33
_upslope = (1, 0, 0, 0, 0.34, 0, 0)
_downslope = (0, 1, 0, 0, 0.34, 0, 0)
_downstair = (0, 0, 0, 1, 0, 0.08, 0.25)
_plane = (0,) * 7
TASK_ENV_VECS = {                                    # :34-41
    "stairslope": (_upstair, _downslope, _plane) * 6,
    "slopestair": (_upslope, _downstair, _plane) * 6,
    "stairstair": (_upstair, _downstair, _plane) * 6,
    "slopeslope": (_upslope, _downslope, _plane) * 6,
}
TASKS = ("plane", "stairslope", "stairstair", "slopestair", "slopeslope", "gallop", "cave", "balancebeam", "highstair")
# the reference's tenth task, "heightfield" (a PyBullet GEOM_HEIGHTFIELD from a pybullet_data PNG), is refused by name: task_terrain()
_IDENTITY = [0.0, 0.0, 0.0, 1]


class _World(object):
    def __init__(self):
        self.boxes = []

    def box(self, half_extents, position, quaternion=_IDENTITY, friction=FRICTION):
        self.boxes.append(Box([float(v) for v in half_extents], [float(v) for v in position], [float(v) for v in quaternion],
                              float(friction)))
        return len(self.boxes) - 1

    def set_friction(self, index, friction):
        self.boxes[index] = self.boxes[index]._replace(friction=float(friction))


def _subplane(w, basex, basez, endx):                                   # terrain.py:174-184
    half = (endx - basex) / 2.0
    if basez <= 0.01:
        basez = 0.01
    w.box([half, 2.5, 0.01], [basex + half, 0, basez - 0.01])
    return endx, basez


def _substair(w, basex, basez, stepwidth, stepheight, stepnum, up):     # :187-204
    for i in range(stepnum):
        z = basez + (i + 0.5) * stepheight if up else basez - (i + 0.5) * stepheight
        w.box([stepwidth / 2, 2.5, stepheight / 2], [basex + (i + 0.5) * stepwidth, 0, z])
    return basex + stepnum * stepwidth, (basez + stepnum * stepheight if up else basez - stepnum * stepheight)


def _subslope(w, basex, basez, slope, endx):                            # :206-218
    half = abs((endx - basex) / np.cos(slope)) / 2.0
    w.box([half, 2.5, 0.01], [basex + half * np.cos(slope), 0, basez + half * np.sin(slope)],
          [0, np.sin(-slope / 2.0), 0, np.cos(-slope / 2.0)])
    return endx, basez + np.sin(slope) * half * 2


def _axis_angle_quaternion(axis, angle):                                # pose3d.QuaternionFromAxisAngle (pose3d.py:97-124)
    q = np.zeros(4, dtype=np.float64)
    q[0:3] = axis
    q[0:3] *= math.sin(angle * 0.5) / np.linalg.norm(axis)
    q[3] = math.cos(angle * 0.5)
    return q


def cal_basez(env_vectors, stepnum, deltax):                            # :131-145
    endz = min_z = 0
    for v in env_vectors:
        if v[0]:
            endz += np.tan(v[4]) * deltax
        elif v[1]:
            endz -= np.tan(v[4]) * deltax
        elif v[2]:
            endz += stepnum * v[5]
        elif v[3]:
            endz -= stepnum * v[5]
        if min_z > endz:
            min_z = endz
    return min_z


def generate_env_vec(mode, num, rng=np.random):                         # :147-171 (same draw order as the reference)
    heads = {"upslope": 0, "downslope": 1, "upstair": 2, "downstair": 3}
    for prefix, head in heads.items():
        if mode.startswith(prefix):
            env_heads = np.array([head] * num)
            break
    else:
        env_heads = rng.choice(4, num)
    stepheights = rng.choice(STEP_HEIGHT, num)
    stepwidths = rng.choice(STEP_WIDTH, num)
    slopes = rng.choice(SLOPE, num)
    vecs = []
    for k in range(num):
        v = np.zeros(7)
        v[env_heads[k]] = 1
        v[4], v[5], v[6] = slopes[k], stepheights[k], stepwidths[k]
        vecs.append(v)
    return vecs


def upstair_terrain(stepwidth=0.33, stepheight=0.05, slope=0.05, stepnum=40, mode="terrain-fix", env_vecs=(), rng=np.random):
    """terrain.py:15-129. Returns (add_height, env_info, boxes)."""
    w, add_height, env_info = _World(), 0, []
    if mode == "stair-fix":                                             # :20-32
        for i in range(stepnum):
            w.box([stepwidth, 2.5, stepheight], [0.66 + i * stepwidth, 0, -stepheight + (i + 1) * stepheight])
        env_info.append([-10, 100, np.array([0, 0, 1, 0, 0, stepheight, stepwidth])])
    elif mode == "stair-var":                                           # :33-47
        basez = 0
        for i in range(5):
            sth = stepheight + i * 0.01
            for j in range(8):
                w.box([stepwidth, 2.5, sth], [0.66 + (i * 8 + j) * stepwidth, 0, basez + j * sth], friction=1.0)
            basez = basez + 8 * sth
    elif mode == "downstair":                                           # :48-51, :221-235
        basez = stepheight * 40
        for i in range(40):
            w.box([2.5, 2.5, stepheight], [0.5 - 2.5 + i * stepwidth, 0, basez - (i + 1) * stepheight])
        add_height = stepheight * stepnum
        env_info.append([-10, 100, np.array([0, 0, 0, 1, 0, stepheight, stepwidth])])
    elif mode == "slope":                                               # :52-58, :237-250
        q = _axis_angle_quaternion([0, 1, 0], -slope)
        if slope > 0:
            w.box([50, 2.5, 0.01], [0.15 + 50 * np.cos(slope), 0, 50 * np.sin(slope)], q)
        else:
            w.box([50, 2.5, 0.01], [-1 + 50 * np.cos(slope), 0, -50 * np.sin(slope)], q)
        if slope < 0:
            add_height = 100 * np.sin(abs(slope)) - np.tan(abs(slope)) + abs(slope) * 0.15
            env_info.append([-10, 100, np.array([0, 1, 0, 0, slope, 0, 0])])
        else:
            env_info.append([-10, 100, np.array([1, 0, 0, 0, slope, 0, 0])])
    elif mode.endswith("random") or mode == "special":                  # :59-117
        vecs = ([np.array(v, dtype=np.float64) for v in env_vecs] if mode == "special" else generate_env_vec(mode, 10, rng))
        deltaz = cal_basez(vecs, STEP_PER_NUM, DELTA_X)
        basex, basez = -1, 0
        if deltaz < 0:
            add_height = basez = abs(deltaz)
        last_x = basex
        basex, basez = _subplane(w, basex, basez, 0.5)
        env_info.append([last_x, basex, np.array([0, 0, 0, 0, 0, 0, 0])])
        for i, v in enumerate(vecs):
            last_x = basex
            if v[0]:
                v[5] = v[6] = 0
                basex, basez = _subslope(w, basex, basez, v[4], basex + DELTA_X)
            elif v[1]:
                v[5] = v[6] = 0
                basex, basez = _subslope(w, basex, basez, -v[4], basex + DELTA_X)
            elif v[2]:
                v[4] = 0
                basex, basez = _substair(w, basex, basez, v[6], v[5], STEP_PER_NUM, True)
            elif v[3]:
                v[4] = 0
                basex, basez = _substair(w, basex, basez, v[6], v[5], STEP_PER_NUM, False)
            else:
                v = np.zeros(7)
                basex, basez = _subplane(w, basex, basez, basex + 1)
            env_info.append([last_x, basex, v])
            last_x = basex
            if i < len(vecs) - 1:                                       # a flat landing between some transitions (:99-108)
                nxt = vecs[i + 1]
                if v[3] and (nxt[0] or nxt[2]):
                    basex, basez = _subplane(w, basex, basez, basex + 0.5)
                elif v[0] and (nxt[1] or nxt[2]):
                    basex, basez = _subplane(w, basex, basez, basex + 0.5)
                elif v[1] and (nxt[2] or nxt[0]):
                    basex, basez = _subplane(w, basex, basez, basex + 0.5)
                elif v[2] and (nxt[0] or nxt[1]):
                    basex, basez = _subplane(w, basex, basez, basex + 0.2)
            if last_x != basex:
                env_info.append([last_x, basex, np.array([0, 0, 0, 0, 0, 0, 0])])
                last_x = basex
        if basez > 0:                                                   # ramp back down to the ground (:112-115)
            deltax = basez / np.tan(0.4)
            basex, basez = _subslope(w, basex, basez, -0.4, basex + deltax)
            env_info.append([last_x, basex, np.array([0, 1, 0, 0, -0.4, 0, 0])])
    elif mode == "balance_beam":                                        # :118-121, :252-266
        add_height = 5
        front = w.box([2.5, 2.5, 0.01], [-2.3, 0, 5], friction=DEFAULT_FRICTION)
        w.set_friction(front, FRICTION)
        w.box([2.5, 2.5, 0.01], [stepheight + 2.7, 0, 5], friction=DEFAULT_FRICTION)
        w.set_friction(front, FRICTION)                                 # the reference's slip: id_front again, not id_end
        w.box([stepheight / 2.0, stepwidth, 0.01], [stepheight / 2.0 + 0.2, 0, 5])
        env_info.append([-10, 100, np.zeros(7)])
    elif mode == "gallop":                                              # :122-125, :268-288
        add_height = 5
        w.box([2.5, 2.5, 0.01], [-2.2, 0, 5])
        current_x = 0.3
        for _ in range(30):
            w.box([0.5 / 2.0, 2.5, 0.01], [current_x + stepwidth + 0.5 / 2.0, 0, 5])
            current_x += stepwidth + 0.5
        w.box([2.5, 2.5, 0.01], [current_x + 2.5 / 2.0, 0, 5])
        env_info.append([-10, 100, np.zeros(7)])
    elif mode == "hurdle":                                              # :126-128, :290-301
        current_x = stepwidth / 2.0
        for _ in range(30):
            w.box([0.01, 4, stepheight / 2.0], [current_x, 0, stepheight / 2.0])
            current_x += stepwidth
        env_info.append([-10, 100, np.zeros(7)])
    elif mode == "cave":                                                # :129-131, :303-316
        w.box([10, stepwidth, 0.01], [10 + 0.3, 0, stepheight])
        w.box([15, 0.01, stepheight / 2.0], [10, stepwidth, stepheight / 2.0])
        w.box([15, 0.01, stepheight / 4.0], [10, -stepwidth, stepheight / 4.0])
        env_info.append([-10, 100, np.zeros(7)])
    return add_height, env_info, w.boxes


# every `task=` that builds a box course (locomotion_gym_env.py:309-325); "plane" builds none
TASK_NAMES = tuple(t for t in TASKS if t != "plane")


def task_terrain(task):
    """What LocomotionGymEnv.reset builds on its first reset for `task_mode=task` (locomotion_gym_env.py:309-325): returns
    (add_height, env_info, boxes). "plane" adds nothing to the ground plane: env_info stays the constructor's single
    up-slope-of-angle-0 stretch (:76). "heightfield" (a PyBullet GEOM_HEIGHTFIELD from a pybullet_data PNG, :160-164) is not a box
    course and is refused by name — A1GymEnv does the same before it gets here."""
    if task == "heightfield":
        raise NotImplementedError("task 'heightfield' is PyBullet's height field (envs/utilities/heightfield.py:89-104): not built")
    if task in TASK_ENV_VECS:
        return upstair_terrain(mode="special", env_vecs=TASK_ENV_VECS[task])
    if task == "gallop":
        return upstair_terrain(stepwidth=0.5, mode="gallop")
    if task == "cave":
        return upstair_terrain(stepheight=0.18, mode="cave")
    if task == "balancebeam":
        return upstair_terrain(stepwidth=0.05, stepheight=6, mode="balance_beam")
    if task == "highstair":
        return upstair_terrain(stepwidth=0.4, stepheight=0.13, mode="stair-fix")
    return 0, [[-100, 100, np.array([1, 0, 0, 0, 0, 0, 0])]], []
