"""`pybullet_utils.transformations` (Gohlke's transformations.py) — only what quadrupedal/envs/utilities/pose3d.py
touches, quaternions as (x, y, z, w). TEST INFRASTRUCTURE, not on any path the goldens record."""
import numpy as np


def quaternion_inverse(q):
    q = np.array(q, dtype=np.float64)
    q[:3] = -q[:3]
    return q / np.dot(q, q)


def quaternion_multiply(q1, q0):
    x0, y0, z0, w0 = q0
    x1, y1, z1, w1 = q1
    return np.array((x1 * w0 + y1 * z0 - z1 * y0 + w1 * x0, -x1 * z0 + y1 * w0 + z1 * x0 + w1 * y0,
                     x1 * y0 - y1 * x0 + z1 * w0 + w1 * z0, -x1 * x0 - y1 * y0 - z1 * z0 + w1 * w0), dtype=np.float64)
