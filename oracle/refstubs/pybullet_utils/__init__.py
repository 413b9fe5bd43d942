"""Stand-in for pybullet_utils (see ../pybullet/__init__.py). TEST INFRASTRUCTURE."""
