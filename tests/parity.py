"""Error metrics shared by the parity tests.

"Within 1e-5 relative" (BASELINE.json north_star) is measured per *physical vector*: the error of a
component is divided by the largest magnitude in the vector it belongs to (a body-frame velocity
triple, a rotation matrix, ...), not by the component itself — a 1e-7 absolute error on a component
that happens to cross zero is not a 100 % error of the simulation.
"""
import numpy as np

REL_TOL = 1e-5  # the north-star tolerance for floating-point dynamics

# obs layout env.py:193-209: b_v(3) b_pos(3) acc(3) gyro(3) pitch/roll/yaw(3) z(1)
OBS_GROUPS = [(0, 3), (3, 6), (6, 9), (9, 12), (12, 15), (15, 16)]


def vec_rel_err(a, b, floor=1e-6):
    """max over rows of |a-b| / (max_j |b_j| + floor); a, b: [..., k]."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    scale = np.max(np.abs(b), axis=-1, keepdims=True) + floor
    return float(np.max(np.abs(a - b) / scale)) if a.size else 0.0


def obs_rel_err(a, b, z_offset=5.0):
    """obs[15] = z + z_offset (env.py:203-204) cancels to ~0 near the floor; its natural scale is
    the offset itself, so that entry is normalised by |z + z_offset| + z_offset."""
    err = max(vec_rel_err(a[..., lo:hi], b[..., lo:hi]) for lo, hi in OBS_GROUPS[:-1])
    return max(err, vec_rel_err(a[..., 15:16], b[..., 15:16], floor=z_offset))


def scalar_rel_err(a, b, floor=1.0):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / (np.abs(b) + floor))) if a.size else 0.0
