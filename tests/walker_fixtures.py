"""Load the parsed MetaLocomotion models committed under tests/golden/walker_models.npz."""
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
