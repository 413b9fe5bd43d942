"""Stand-in for the `pybullet_data` package (TEST INFRASTRUCTURE, build container only): the quadrupedal reference
only asks it where its asset directory is (a1/a1.urdf, plane_implicit.urdf live there — and are NOT in the
reference tree, which is why the A1 body itself cannot be pinned)."""
import os


def getDataPath():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
