import numpy as np


def np_random(seed=None):
    rng = np.random.RandomState(seed)
    return rng, seed
