#!/usr/bin/env python3
"""Parse a handful of the reference's MetaLocomotion MJCF assets into flat model arrays
(tests/golden/walker_models.npz) so the GPU box — which has no /root/reference — can run the
walker tests on the real robots. TEST INFRASTRUCTURE; run in the build container:

    python oracle/gen_golden_walker.py

Only numbers derived by metagym_amd.metalocomotion.mjcf.load_mjcf are stored (topology, frames,
inertias, joint parameters, collision spheres) — not the XML text.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = os.environ.get("METAGYM_REFERENCE", "/root/reference")
ASSETS = os.path.join(REF, "metagym", "metalocomotion", "envs", "assets")

from metagym_amd.metalocomotion.mjcf import load_mjcf  # noqa: E402

ANT_FEET = ("front_left_foot", "front_right_foot", "left_back_foot", "right_back_foot")
FILES = [("humanoid", "humanoids/humanoid.xml", ("right_foot", "left_foot")),
         ("humanoid_tra_000", "humanoids/humanoid_var_tra_000.xml", ("right_foot", "left_foot")),
         ("humanoid_tra_137", "humanoids/humanoid_var_tra_137.xml", ("right_foot", "left_foot")),
         ("humanoid_ood_003", "humanoids/humanoid_var_ood_003.xml", ("right_foot", "left_foot")),
         ("ant", "ants/ant.xml", ANT_FEET),
         ("ant_tra_005", "ants/ant_var_tra_005.xml", ANT_FEET)]

out = {}
for key, rel, feet in FILES:
    # "<key>": the loader's default preset ("bullet", what the envs run by default); "<key>@mujoco": MuJoCo's reading
    for name, preset in ((key, None), (key + "@mujoco", "mujoco")):
        m = load_mjcf(os.path.join(ASSETS, rel), foot_names=feet, preset=preset)
        for k, v in m.to_dict().items():
            out["%s/%s" % (name, k)] = v
        print(name, str(m.preset), "bodies", len(m.body_parent), "joints", len(m.joint_body), "mass %.3f" % m.body_mass.sum(),
              "inertia trace body 0 %.4f" % np.trace(m.body_inertia[0]))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "walker_models.npz"), **out)
