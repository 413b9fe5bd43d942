"""Load the parsed MetaLocomotion models committed under tests/golden/walker_models.npz: "<name>" read with the loader's
default preset ("bullet", what the envs run by default), "<name>@mujoco" with MuJoCo's reading of the same file."""
import os

import numpy as np

from metagym_amd.metalocomotion.mjcf import Model

_PATH = os.path.join(os.path.dirname(__file__), "golden", "walker_models.npz")


def load_models():
    z = np.load(_PATH)
    groups = {}
    for k in z.files:
        name, field = k.split("/", 1)
        groups.setdefault(name, {})[field] = z[k]
    return {name: Model.from_dict(d) for name, d in groups.items()}


def world_kw(m):
    """The world half of the preset a Model was loaded with (mjcf.PRESETS) as oracle/abd.Params keywords: btMultiBody's body
    velocity damping, its clamp of the generalized velocities and the contact margin ("bullet": 0.04 / 0.04, 100 and 0.02 m;
    "mujoco": off)."""
    bd = getattr(m, "body_damping", (0.0, 0.0))
    return dict(body_damping=(float(bd[0]), float(bd[1])), max_velocity=float(getattr(m, "max_velocity", 0.0)),
                contact_margin=margin_of(m))


def margin_of(m):
    """The per-proxy contact margins of the preset's world (mjcf.PRESETS[...]["contact_margin"]: Bullet's relative rule — 0.02 x
    the link's angular motion disc — in the "bullet" world, 0 in the "mujoco" one). The recorded models predate the field and
    carry the preset's name only."""
    from metagym_amd.metalocomotion import mjcf
    rule = getattr(m, "contact_margin", None)
    if rule is None:
        rule = mjcf.PRESETS[preset_of(m)]["contact_margin"]
    rule = rule.item() if hasattr(rule, "item") else rule
    return mjcf.contact_margins(m, rule if isinstance(rule, str) else float(rule))


def preset_of(m):
    return str(getattr(m, "preset", "mujoco"))
