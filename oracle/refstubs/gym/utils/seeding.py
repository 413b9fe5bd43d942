import numpy as np

# The reference reseeds from OS entropy on every reset (env_bases.py:57-66 -> np_random(None)). Golden generation
# needs repeatable draws: set FORCED_SEEDS to an iterator of seeds and np_random(None) consumes one per call.
FORCED_SEEDS = None


def np_random(seed=None):
    if seed is None and FORCED_SEEDS is not None:
        seed = next(FORCED_SEEDS)
    rng = np.random.RandomState(seed)
    return rng, seed
