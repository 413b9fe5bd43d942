"""URDF test fixtures (tests only; neither is the reference's robot file — `a1/a1.urdf` ships with pybullet_data and is not
in the reference tree):

  model_to_urdf(m)   writes a `Model` (e.g. the MJCF stand-in body of examples/a1_standin) as URDF text: explicit
                     `<inertial>`, capsule collisions, `<contact>` friction, one revolute joint per body. Read back with
                     `load_urdf(..., inertia="file")` it must give the SAME arrays, bit for bit (tests/test_urdf.py), which is
                     what lets tests/test_a1_physics_gpu.py demand URDF path == MJCF path bit for bit on the engine.
  a1_like_urdf()     an A1-SHAPED robot written from the kinematic constants the reference's own Python states (hip offsets
                     and the 0.08505 / 0.2 / 0.2 m leg segments of robots/a1.py:61-63,88-123, joint names and ranges of
                     a1.py:27-40,158-196) with made-up inertial data: box trunk, cylinder hips, box thighs / calves, sphere
                     toes, and the fixed joints such files carry (imu, a hip rotor, toes) — everything the loader has to handle.
"""
import numpy as np


def _f(x):
    return " ".join(repr(float(v)) for v in np.atleast_1d(x))


def _capsule_origin(p0, p1):
    """(xyz, rpy, length) of a capsule whose local z axis runs from p0 to p1 (axis-aligned capsules only: exact rpy)."""
    p0, p1 = np.asarray(p0, float), np.asarray(p1, float)
    d = p1 - p0
    L = float(np.linalg.norm(d))
    if L == 0.0:
        return p0, (0.0, 0.0, 0.0), 0.0
    k = int(np.argmax(np.abs(d)))
    assert np.count_nonzero(d) == 1, "axis-aligned capsules only"
    s = 1.0 if d[k] > 0 else -1.0
    hp = np.pi / 2
    rpy = {(0, 1.0): (0.0, hp, 0.0), (0, -1.0): (0.0, -hp, 0.0), (1, 1.0): (-hp, 0.0, 0.0), (1, -1.0): (hp, 0.0, 0.0),
           (2, 1.0): (0.0, 0.0, 0.0), (2, -1.0): (np.pi, 0.0, 0.0)}[(k, s)]
    return 0.5 * (p0 + p1), rpy, L


def model_to_urdf(m, name="model"):
    nb = len(m.body_parent)
    assert np.all(np.asarray(m.joint_anchor) == 0.0) and all(np.array_equal(R, np.eye(3)) for R in m.body_rot)
    out = ['<?xml version="1.0"?>', '<robot name="%s">' % name]
    for b in range(nb):
        out.append('  <link name="%s">' % m.body_names[b])
        I = m.body_inertia[b]
        out.append('    <inertial><origin xyz="%s" rpy="0 0 0"/><mass value="%s"/>' % (_f(m.body_com[b]), _f(m.body_mass[b])))
        out.append('      <inertia ixx="%s" ixy="%s" ixz="%s" iyy="%s" iyz="%s" izz="%s"/></inertial>'
                   % (_f(I[0, 0]), _f(I[0, 1]), _f(I[0, 2]), _f(I[1, 1]), _f(I[1, 2]), _f(I[2, 2])))
        for g in np.nonzero(np.asarray(m.geom_body) == b)[0]:
            xyz, rpy, L = _capsule_origin(m.geom_p0[g], m.geom_p1[g])
            geo = '<sphere radius="%s"/>' % _f(m.geom_radius[g]) if L == 0.0 else \
                '<capsule radius="%s" length="%s"/>' % (_f(m.geom_radius[g]), _f(L))
            out.append('    <collision><origin xyz="%s" rpy="%s"/><geometry>%s</geometry></collision>' % (_f(xyz), _f(rpy), geo))
        out.append('    <contact><lateral_friction value="%s"/></contact>' % _f(m.geom_friction))
        out.append('  </link>')
    for j, b in enumerate(m.joint_body):
        out.append('  <joint name="%s" type="revolute">' % m.joint_names[j])
        out.append('    <parent link="%s"/><child link="%s"/>' % (m.body_names[m.body_parent[b]], m.body_names[b]))
        out.append('    <origin xyz="%s" rpy="0 0 0"/><axis xyz="%s"/>' % (_f(m.body_pos[b]), _f(m.joint_axis[j])))
        out.append('    <limit lower="%s" upper="%s" effort="33.5" velocity="21"/><dynamics damping="%s" friction="0"/>'
                   % (_f(m.joint_lo[j]), _f(m.joint_hi[j]), _f(m.joint_damping[j])))
        out.append('  </joint>')
    out.append('</robot>')
    return "\n".join(out)


A1_LIKE_LEGS = (("FR", 1, -1), ("FL", 1, 1), ("RR", -1, -1), ("RL", -1, 1))
A1_LIKE_TOES = tuple("%s_toe" % leg for leg, _, _ in A1_LIKE_LEGS)


def a1_like_urdf(shuffle_legs=False):
    """See the module docstring. `shuffle_legs` writes the legs in RL, FR, RR, FL document order (the loader must still
    deliver the hinges in a1.MOTOR_NAMES order when asked to)."""
    box_inertia = lambda m, s: (m / 12.0 * (s[1] ** 2 + s[2] ** 2), m / 12.0 * (s[0] ** 2 + s[2] ** 2), m / 12.0 * (s[0] ** 2 + s[1] ** 2))
    o = ['<?xml version="1.0"?>', '<robot name="a1_like_test_fixture">']

    def link(name, mass, com, diag, cols, friction=None, rpy="0 0 0", offdiag=(0.0, 0.0, 0.0)):
        o.append('  <link name="%s">' % name)
        o.append('    <inertial><origin xyz="%s" rpy="%s"/><mass value="%s"/><inertia ixx="%s" ixy="%s" ixz="%s" iyy="%s" iyz="%s" izz="%s"/></inertial>'
                 % (_f(com), rpy, _f(mass), _f(diag[0]), _f(offdiag[0]), _f(offdiag[1]), _f(diag[1]), _f(offdiag[2]), _f(diag[2])))
        for xyz, crpy, geo in cols:
            o.append('    <collision><origin xyz="%s" rpy="%s"/><geometry>%s</geometry></collision>' % (_f(xyz), crpy, geo))
        if friction is not None:
            o.append('    <contact><lateral_friction value="%s"/></contact>' % _f(friction))
        o.append('  </link>')

    def joint(name, kind, parent, child, xyz, axis=None, lim=None, rpy="0 0 0"):
        o.append('  <joint name="%s" type="%s"><parent link="%s"/><child link="%s"/><origin xyz="%s" rpy="%s"/>' % (name, kind, parent, child, _f(xyz), rpy))
        if axis is not None:
            o.append('    <axis xyz="%s"/><limit lower="%s" upper="%s" effort="33.5" velocity="21"/><dynamics damping="0.01" friction="0"/>'
                     % (_f(axis), _f(lim[0]), _f(lim[1])))
        o.append('  </joint>')

    trunk = (0.267, 0.194, 0.114)
    link("trunk", 4.7, (0.0127, 0.0022, 0.0005), box_inertia(4.7, trunk), [((0, 0, 0), "0 0 0", '<box size="%s"/>' % _f(trunk))],
         offdiag=(3e-5, 1.1e-4, 1e-6))
    link("imu_link", 0.001, (0, 0, 0), (1e-4, 1e-6, 1e-4), [])
    joint("imu_joint", "fixed", "trunk", "imu_link", (0, 0, 0))
    legs = A1_LIKE_LEGS if not shuffle_legs else (A1_LIKE_LEGS[3], A1_LIKE_LEGS[0], A1_LIKE_LEGS[2], A1_LIKE_LEGS[1])
    for leg, sx, sy in legs:
        hip_cyl = '<cylinder radius="0.046" length="0.04"/>'
        link(leg + "_hip", 0.7, (-0.003 * sx, -0.001 * sy, 0.0), (4.7e-4, 8.1e-4, 5.5e-4), [((0, 0, 0), "1.5707963267948966 0 0", hip_cyl)])
        joint(leg + "_hip_joint", "revolute", "trunk", leg + "_hip", (0.183 * sx, 0.047 * sy, 0.0), (1, 0, 0), (-0.802851455917, 0.802851455917))
        link(leg + "_upper_shoulder", 0.05, (0, 0, 0), (1e-5, 1e-5, 1e-5), [((0, 0, 0), "1.5708 0 0", '<cylinder radius="0.041" length="0.032"/>')])
        joint(leg + "_hip_fixed", "fixed", leg + "_hip", leg + "_upper_shoulder", (0, 0.081 * sy, 0))
        thigh = (0.2, 0.0245, 0.034)
        link(leg + "_upper", 1.0, (-0.0032, -0.0223 * sy, -0.0279), (5.5e-3, 5.1e-3, 1.4e-3),
             [((0, 0, -0.1), "0 1.5707963267948966 0", '<box size="%s"/>' % _f(thigh))], offdiag=(-4e-6 * sy, 3.4e-4, 2e-5 * sy))
        joint(leg + "_upper_joint", "revolute", leg + "_hip", leg + "_upper", (0, 0.08505 * sy, 0), (0, 1, 0), (-1.0471975512, 4.18879020479))
        calf = (0.2, 0.016, 0.016)
        link(leg + "_lower", 0.17, (0.0065, 0, -0.1073), (3.0e-3, 3.0e-3, 3.2e-5), [((0, 0, -0.1), "0 1.5707963267948966 0", '<box size="%s"/>' % _f(calf))])
        joint(leg + "_lower_joint", "revolute", leg + "_upper", leg + "_lower", (0, 0, -0.2), (0, 1, 0), (-2.69653369433, -0.916297857297))
        link(leg + "_toe", 0.06, (0, 0, 0), (9.6e-6, 9.6e-6, 9.6e-6), [((0, 0, 0), "0 0 0", '<sphere radius="0.02"/>')], friction=0.4)
        joint(leg + "_toe_fixed", "fixed", leg + "_lower", leg + "_toe", (0, 0, -0.2))
    o.append('</robot>')
    return "\n".join(o)


if __name__ == "__main__":      # python tests/urdf_fixture.py examples/a1_like/a1_like.urdf
    import sys
    header = ("<!-- A1-SHAPED TEST / DEMO ROBOT, written by tests/urdf_fixture.py:a1_like_urdf(). NOT the reference's robot file: that is\n"
              "     pybullet_data/a1/a1.urdf, which is not in the reference tree. Leg geometry and joint ranges are the constants the\n"
              "     reference's own Python states (robots/a1.py:61-63,88-123,158-196); masses, inertias and collision shapes are made up. -->\n")
    text = a1_like_urdf()
    first, rest = text.split("\n", 1)
    open(sys.argv[1], "w").write(first + "\n" + header + rest + "\n")
