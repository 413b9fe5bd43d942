#!/usr/bin/env python3
"""Golden vectors for the quadrupedal terrains: the UNMODIFIED reference terrain builder on a recording `pybullet`.

TEST INFRASTRUCTURE; runs only in the build container (needs /root/reference):

    python oracle/gen_golden_a1_terrain.py

metagym/quadrupedal/envs/utilities/terrain.py creates its ground as static boxes through four PyBullet calls
(createCollisionShape, createMultiBody, changeDynamics, GEOM_BOX). Here the module's `p` is replaced by a recorder, so
each case yields exactly what the reference would have put into its world — per body: half extents, position, orientation
and the lateral friction it ends up with (NaN = changeDynamics never called on it) — plus the returned `add_height` and
`env_info`. Cases: the eight terrains LocomotionGymEnv.reset builds from `task_mode` (locomotion_gym_env.py:309-325, with
that module's own env-vector lists), every `hardset` mode with non-default parameters (:298-301), and the random modes
under `np.random.seed(s)`. metagym_amd/quadrupedal/terrain.py must reproduce all of it bit for bit
(tests/test_a1_terrain.py)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import gen_golden_a1 as ga  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "a1_terrain.npz")


class Recorder(object):
    GEOM_BOX = 3

    def __init__(self):
        self.shapes, self.bodies = [], []

    def createCollisionShape(self, kind, halfExtents):
        assert kind == self.GEOM_BOX
        self.shapes.append([float(v) for v in halfExtents])
        return len(self.shapes) - 1

    def createMultiBody(self, baseMass, baseCollisionShapeIndex, basePosition, baseOrientation):
        assert baseMass == 0
        self.bodies.append(self.shapes[baseCollisionShapeIndex] + [float(v) for v in basePosition] +
                           [float(v) for v in baseOrientation] + [float("nan")])
        return len(self.bodies) - 1

    def changeDynamics(self, body, link, lateralFriction):
        assert link == -1
        self.bodies[body][10] = float(lateralFriction)


def main():
    ga.import_reference()
    from metagym.quadrupedal.envs.utilities import terrain
    from metagym.quadrupedal.envs import locomotion_gym_env as lge

    def run(seed=None, **kw):
        terrain.p = rec = Recorder()
        if seed is not None:
            np.random.seed(seed)
        if "env_vecs" in kw:
            kw["env_vecs"] = [np.array(v, dtype=np.float64) for v in kw["env_vecs"]]
        add_height, env_info = terrain.upstair_terrain(**kw)
        rows = np.array([[r[0], r[1]] + [float(x) for x in r[2]] for r in env_info], dtype=np.float64).reshape(len(env_info), 9)
        return float(add_height), rows, np.array(rec.bodies, dtype=np.float64).reshape(len(rec.bodies), 11)

    cases = []
    # the task table of LocomotionGymEnv.reset (locomotion_gym_env.py:309-325)
    table = {"stairslope": dict(mode="special", env_vecs=lge.upstair_downslope),
             "stairstair": dict(mode="special", env_vecs=lge.upstair_downstair),
             "slopestair": dict(mode="special", env_vecs=lge.upslope_downstair),
             "slopeslope": dict(mode="special", env_vecs=lge.upslope_downslope),
             "gallop": dict(stepwidth=0.5, mode="gallop"),
             "cave": dict(stepheight=0.18, mode="cave"),
             "balancebeam": dict(stepwidth=0.05, stepheight=6, mode="balance_beam"),
             "highstair": dict(stepwidth=0.4, stepheight=0.13, mode="stair-fix")}
    for task, kw in table.items():
        cases.append(("task:" + task, None, kw))
    # hardset modes (:298-301) away from their defaults
    for kw in (dict(mode="stair-fix", stepwidth=0.31, stepheight=0.07, stepnum=12), dict(mode="stair-var", stepwidth=0.3, stepheight=0.04),
               dict(mode="downstair", stepwidth=0.29, stepheight=0.06, stepnum=25), dict(mode="slope", slope=0.23),
               dict(mode="slope", slope=-0.31), dict(mode="hurdle", stepwidth=0.7, stepheight=0.15),
               dict(mode="cave", stepwidth=0.3, stepheight=0.33), dict(mode="gallop", stepwidth=0.35),
               dict(mode="balance_beam", stepwidth=0.08, stepheight=4.0), dict(mode="terrain-fix"),
               dict(mode="special", env_vecs=[[0, 1, 0, 0, 0.3, 0.1, 0.3], [0, 0, 1, 0, 0.2, 0.09, 0.28], [1, 0, 0, 0, 0.4, 0.1, 0.3],
                                              [0, 0, 0, 1, 0.2, 0.1, 0.3], [0, 0, 1, 0, 0.1, 0.08, 0.3], [0, 1, 0, 0, 0.45, 0, 0],
                                              [1, 0, 0, 0, 0.2, 0, 0], [0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 1, 0.0, 0.1, 0.26]])):
        cases.append(("hardset:%s:%d" % (kw["mode"], len(cases)), None, kw))
    for seed, mode in enumerate(["random", "random", "random", "upstair-random", "downstair-random", "upslope-random",
                                 "downslope-random", "random"]):
        cases.append(("random:%s:%d" % (mode, seed), 100 + seed, dict(mode=mode)))

    out, meta = {}, []
    for name, seed, kw in cases:
        spec = {k: ([[float(x) for x in v] for v in val] if k == "env_vecs" else val) for k, val in kw.items()}
        add_height, env_info, bodies = run(seed, **dict(kw))
        out[name + "/add_height"], out[name + "/env_info"], out[name + "/bodies"] = np.float64(add_height), env_info, bodies
        meta.append(dict(name=name, seed=seed, kwargs=spec))
        print("%-28s add_height %-8.5g env_info %2d rows, %3d boxes" % (name, add_height, len(env_info), len(bodies)))
    out["cases"] = np.array(json.dumps(meta))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
