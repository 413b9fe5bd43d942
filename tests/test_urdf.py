"""Host side of SURVEY.md §8(f)-2, CPU only: the URDF loader (metagym_amd/quadrupedal/urdf.py), the loader options that
restate what PyBullet's importers are known to do (bounding-box inertia, body velocity damping), and the numpy engine
(oracle/abd.py) on a URDF robot. The fixtures are written by tests/urdf_fixture.py — neither is the reference's a1.urdf
(pybullet_data, absent from the reference tree); dynamics parity with PyBullet stays unpinned."""
import copy
import os

import numpy as np
import pytest

from metagym_amd.metalocomotion import variants
from metagym_amd.metalocomotion.mjcf import load_mjcf
from metagym_amd.quadrupedal import MOTOR_NAMES
from metagym_amd.quadrupedal.urdf import load_urdf, rpy_to_mat
from oracle import abd
from urdf_fixture import A1_LIKE_TOES, a1_like_urdf, model_to_urdf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STANDIN_XML = os.path.join(ROOT, "examples", "a1_standin", "a1_standin.xml")
CALVES = ("FR_calf", "FL_calf", "RR_calf", "RL_calf")
MODEL_ARRAYS = ("body_parent", "body_pos", "body_rot", "body_mass", "body_com", "body_inertia", "joint_body", "joint_anchor",
                "joint_axis", "joint_lo", "joint_hi", "joint_armature", "joint_damping", "joint_stiffness", "sph_body", "sph_pos",
                "sph_radius", "foot_body")


def test_quarter_turns_are_exact():
    for txt in (1.5708, 1.57079632679, np.pi / 2):
        R = rpy_to_mat([txt, 0, 0])
        assert np.array_equal(R, [[1, 0, 0], [0, 0, -1], [0, 1, 0]])
    assert np.array_equal(rpy_to_mat([0, -np.pi / 2, np.pi]), [[0, 0, 1], [0, -1, 0], [1, 0, 0]])
    R = rpy_to_mat([0.3, -0.2, 1.1])                                     # a general rotation: orthonormal, right-handed
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-15) and np.isclose(np.linalg.det(R), 1.0)


def test_urdf_of_the_standin_body_loads_to_the_same_model_bit_for_bit():
    """MJCF -> Model -> URDF text -> Model: every array identical (what lets the GPU test demand URDF path == MJCF path)."""
    m = load_mjcf(STANDIN_XML, foot_names=CALVES, preset="mujoco")       # (the stand-in's joints carry armature 0.01, like the URDF side)
    u = load_urdf(model_to_urdf(m), foot_links=CALVES, inertia="file", armature=0.01, root_pose=((0, 0, 0.28), None))
    for k in MODEL_ARRAYS:
        a, b = np.asarray(getattr(m, k)), np.asarray(getattr(u, k))
        assert a.shape == b.shape and np.array_equal(a, b), k
    assert list(u.joint_names) == list(m.joint_names) == MOTOR_NAMES
    assert np.all(u.sph_friction == float(m.geom_friction))
    assert [int(f) for f in u.sph_foot] == [next((i for i, fb in enumerate(m.foot_body) if fb == b), -1) for b in m.sph_body]


@pytest.mark.parametrize("shuffle", [False, True])
def test_a1_like_urdf_structure(shuffle):
    m = load_urdf(a1_like_urdf(shuffle_legs=shuffle), foot_links=A1_LIKE_TOES, joint_order=MOTOR_NAMES)
    assert list(m.joint_names) == MOTOR_NAMES                            # a1.py:27-40, whatever the document order
    assert len(m.body_parent) == 13 and len(m.link_names) == 22          # 9 links hang on fixed joints: merged
    assert m.body_names[1:4] == ["FR_hip", "FR_upper", "FR_lower"]
    assert m.link_body["imu_link"] == 0 and m.link_body["FR_upper_shoulder"] == m.link_body["FR_hip"]
    assert m.link_body["FR_toe"] == m.link_body["FR_lower"] and np.allclose(m.link_frame["FR_toe"][1], [0, 0, -0.2])
    assert np.isclose(m.body_mass.sum(), 4.7 + 0.001 + 4 * (0.7 + 0.05 + 1.0 + 0.17 + 0.06))
    # proxies: box 8 corners, cylinder 2 rims x rim_points, sphere 1; within the engine's 128
    assert m.rim_points == 3 and len(m.sph_body) == 8 + 4 * (2 * 2 * 3 + 8 + 8 + 1) == 124
    toes = np.nonzero(m.sph_foot >= 0)[0]
    assert len(toes) == 4 and [int(m.sph_foot[g]) for g in toes] == [0, 1, 2, 3]
    assert np.all(m.sph_radius[toes] == 0.02) and np.all(m.sph_friction[toes] == 0.4) and np.all(np.delete(m.sph_friction, toes) == 0.5)
    assert [int(m.sph_body[g]) for g in toes] == [int(b) for b in m.foot_body] == [3, 6, 9, 12]
    assert np.allclose(m.root_inertial_pos, [0.0127, 0.0022, 0.0005])
    # hip x, upper / lower y axes; child frames on their joints
    assert np.array_equal(m.joint_axis[0], [1, 0, 0]) and np.array_equal(m.joint_axis[1], [0, 1, 0]) and not m.joint_anchor.any()
    assert np.allclose(m.body_pos[1], [0.183, -0.047, 0]) and np.allclose(m.body_pos[2], [0, -0.08505, 0]) and np.allclose(m.body_pos[3], [0, 0, -0.2])


def test_merged_links_give_the_composite_rigid_body():
    """calf + toe welded: mass, centre of mass and inertia of the pair, checked against point-mass bookkeeping."""
    m = load_urdf(a1_like_urdf(), foot_links=A1_LIKE_TOES, inertia="file")
    b = m.link_body["FR_lower"]
    m1, c1, I1 = 0.17, np.array([0.0065, 0, -0.1073]), np.diag([3.0e-3, 3.0e-3, 3.2e-5])
    m2, c2, I2 = 0.06, np.array([0, 0, -0.2]), np.diag([9.6e-6] * 3)
    M = m1 + m2
    c = (m1 * c1 + m2 * c2) / M
    par = lambda mm, d: mm * (d @ d * np.eye(3) - np.outer(d, d))
    I = I1 + par(m1, c1 - c) + I2 + par(m2, c2 - c)
    assert np.isclose(m.body_mass[b], M) and np.allclose(m.body_com[b], c, atol=1e-15) and np.allclose(m.body_inertia[b], I, atol=1e-15)


def test_bullet_aabb_inertia_option():
    """loadURDF without URDF_USE_INERTIA_FROM_FILE (a1.py:266-277): the box formula on the collision shapes' bounding box in
    the inertial frame. A box link: its own solid-box inertia; a cylinder lying along y: the box that encloses it; a link
    without collision shapes: nothing; the file's off-diagonal terms are gone."""
    f = load_urdf(a1_like_urdf(), foot_links=A1_LIKE_TOES, inertia="file")
    a = load_urdf(a1_like_urdf(), foot_links=A1_LIKE_TOES, inertia="bullet_aabb")
    for k in ("body_mass", "body_com", "body_pos", "sph_pos"):           # the option touches inertias only
        assert np.array_equal(getattr(f, k), getattr(a, k)), k
    assert f.body_inertia[0][0, 2] != 0.0
    s, mt = (0.267, 0.194, 0.114), 4.7
    trunk_box = mt / 12.0 * np.array([s[1] ** 2 + s[2] ** 2, s[0] ** 2 + s[2] ** 2, s[0] ** 2 + s[1] ** 2])
    # trunk body = trunk link (box inertia about its own inertial origin) + imu_link (no shapes: mass only)
    c0 = np.array([0.0127, 0.0022, 0.0005])
    com = (mt * c0) / (mt + 0.001)
    par = lambda mm, d: mm * (d @ d * np.eye(3) - np.outer(d, d))
    assert np.allclose(a.body_inertia[0], np.diag(trunk_box) + par(mt, c0 - com) + par(0.001, -com), atol=1e-14)
    # hip link alone (inertia="bullet_aabb" on a cylinder r = 0.046, l = 0.04 along y): box 0.092 x 0.04 x 0.092
    only_hip = load_urdf(a1_like_urdf().replace('<joint name="FR_hip_fixed" type="fixed">', '<joint name="FR_hip_fixed" type="revolute">')
                         .replace('<origin xyz="0.0 -0.081 0.0" rpy="0 0 0"/>', '<origin xyz="0.0 -0.081 0.0" rpy="0 0 0"/><axis xyz="0 1 0"/><limit lower="-1" upper="1"/>', 1),
                         foot_links=A1_LIKE_TOES)
    bh = only_hip.link_body["FR_hip"]
    l = np.array([0.092, 0.04, 0.092])
    assert np.allclose(only_hip.body_inertia[bh], np.diag(0.7 / 12.0 * np.array([l[1] ** 2 + l[2] ** 2, l[0] ** 2 + l[2] ** 2, l[0] ** 2 + l[1] ** 2])), atol=1e-15)


def test_unsupported_urdf_features_are_refused_loudly():
    bad = a1_like_urdf().replace('<sphere radius="0.02"/>', '<mesh filename="toe.stl"/>', 1)
    with pytest.raises(ValueError, match="mesh"):
        load_urdf(bad, foot_links=A1_LIKE_TOES)
    assert len(load_urdf(bad, foot_links=A1_LIKE_TOES, mesh="skip").sph_body) == 123
    with pytest.raises(ValueError, match="prismatic"):
        load_urdf(a1_like_urdf().replace('name="FR_hip_joint" type="revolute"', 'name="FR_hip_joint" type="prismatic"'), foot_links=A1_LIKE_TOES)


# ---------------------------------------------------------------------------------------------- MJCF loader options
def test_mjcf_bullet_box_inertia_changes_inertias_only():
    """mjcf.load_mjcf(inertia='bullet_box'): what robot_bases.py:119's loadMJCF (no URDF_USE_INERTIA_FROM_FILE) makes of a
    body's capsules — masses, centres of mass, frames, joints and collision proxies are untouched, the inertia becomes the
    diagonal bounding-box one (a larger trace than the solid capsules')."""
    g, b = variants.model("humanoid", preset="mujoco"), variants.model("humanoid", preset="mujoco", inertia="bullet_box")
    for k in MODEL_ARRAYS:
        if k != "body_inertia":
            assert np.array_equal(getattr(g, k), getattr(b, k)), k
    for Ig, Ib in zip(g.body_inertia, b.body_inertia):
        assert np.count_nonzero(Ib - np.diag(np.diag(Ib))) == 0
        assert np.trace(Ib) > np.trace(Ig)                               # a solid box around the shapes out-weighs them
    o = variants.model("humanoid", preset="mujoco", com="body_origin")
    assert not o.body_com.any() and np.array_equal(o.body_mass, g.body_mass)


def test_mjcf_presets_are_what_the_table_says():
    """DESIGN.md §3.4's "default" column: the `bullet` preset (the loader's and the envs' default) = bounding-box inertia +
    inertial frame at the body origin + armature, joint damping and stiffness ignored + 0.04 / 0.04 body damping + the +-100
    velocity clamp; `mujoco` = MuJoCo's documented reading. Topology, frames, masses and the collision proxies agree."""
    from metagym_amd.metalocomotion.mjcf import DEFAULT_PRESET, PRESETS
    assert DEFAULT_PRESET == "bullet"
    for robot in ("humanoid", "ant"):
        d, b, mj = variants.model(robot, "TRAIN", 5), variants.model(robot, "TRAIN", 5, preset="bullet"), variants.model(robot, "TRAIN", 5, preset="mujoco")
        assert d is b and str(b.preset) == "bullet" and str(mj.preset) == "mujoco"
        combo = variants.model(robot, "TRAIN", 5, preset="mujoco", inertia="bullet_box", com="body_origin")
        assert np.array_equal(b.body_inertia, combo.body_inertia) and not b.body_com.any()
        assert not b.joint_armature.any() and not b.joint_stiffness.any() and not b.joint_damping.any()
        assert mj.joint_armature.all() and mj.joint_damping.all()
        assert tuple(b.body_damping) == (0.04, 0.04) and float(b.max_velocity) == 100.0
        assert tuple(mj.body_damping) == (0.0, 0.0) and float(mj.max_velocity) == 0.0
        for k in MODEL_ARRAYS:
            if k not in ("body_inertia", "body_com", "joint_armature", "joint_damping", "joint_stiffness"):
                assert np.array_equal(getattr(b, k), getattr(mj, k)), k
    with pytest.raises(ValueError):
        variants.model("humanoid", preset="havok")
    assert set(PRESETS) == {"bullet", "mujoco"}


def _free(m):
    m = copy.deepcopy(m)
    m.joint_damping = np.zeros_like(m.joint_damping)
    m.joint_lo, m.joint_hi = np.full_like(m.joint_lo, -100.0), np.full_like(m.joint_hi, 100.0)
    return m


def _random_state(m, seed, z=10.0):
    rs = np.random.RandomState(seed)
    s = abd.State(m)
    s.pos[2] = z
    nj = len(m.joint_lo)
    s.q, s.qd = rs.uniform(-0.3, 0.3, nj), rs.uniform(-2, 2, nj)
    s.w, s.v = rs.uniform(-1, 1, 3), rs.uniform(-1, 1, 3)
    return s


@pytest.mark.parametrize("inertia", ["file", "bullet_aabb"])
def test_urdf_robot_in_free_flight_conserves_momentum(inertia):
    """Explicit (full, off-diagonal) link inertias and merged links through the numpy engine: without gravity, damping or
    limits the linear and angular momentum of the A1-like robot stay put and the energy drift halves with dt."""
    m = _free(load_urdf(a1_like_urdf(), foot_links=A1_LIKE_TOES, inertia=inertia, joint_order=MOTOR_NAMES))
    for I in m.body_inertia:
        assert np.allclose(I, I.T) and np.all(np.linalg.eigvalsh(I) > 0)
    drifts = []
    for dt in (0.001, 0.0005):
        s = _random_state(m, 0)
        T0, _ = abd.energy(m, s)
        P0, L0 = abd.momentum(m, s)
        prm = abd.Params(dt=dt, self_collision=False, gravity=0.0)
        for _ in range(int(round(0.04 / dt))):
            abd.substep(m, s, np.zeros(12), prm)
        T1, _ = abd.energy(m, s)
        P1, L1 = abd.momentum(m, s)
        drifts.append((abs(T1 - T0) / T0, np.abs(P1 - P0).max() / np.abs(P0).max(), np.abs(L1 - L0).max() / np.abs(L0).max()))
    for a, b in zip(*drifts):
        assert a < 2e-3 and b < 0.6 * a + 1e-9, drifts


def test_body_damping_option_changes_only_the_bias_and_drains_energy():
    """Params.body_damping = btMultiBody's linear / angular damping (PyBullet default 0.04 each): M is untouched, the bias
    gains J^T of (m v (k + k|v|), I w (k + k|w|)) per body, a free-flying robot loses kinetic energy at the rate that wrench
    dissipates, and with (0, 0) nothing changes at all."""
    m = _free(variants.model("humanoid", preset="mujoco"))
    s = _random_state(m, 4)
    M0, h0, kin, _ = abd.mass_matrix_and_bias(m, s, gravity=0.0)
    M1, h1, _, _ = abd.mass_matrix_and_bias(m, s, gravity=0.0, body_damping=(0.04, 0.04))
    assert np.array_equal(M0, M1)
    u = s.u()
    power = 0.0
    expect = np.zeros_like(h0)
    for b in range(len(m.body_parent)):
        Jv, Jw = abd.point_jacobian(m, kin, b, kin["c"][b]), abd.angular_jacobian(m, kin, b)
        vc, w = Jv @ u, Jw @ u
        Iw = kin["R"][b] @ m.body_inertia[b] @ kin["R"][b].T
        F = m.body_mass[b] * (0.04 + 0.04 * np.linalg.norm(vc)) * vc
        N = (0.04 + 0.04 * np.linalg.norm(w)) * (Iw @ w)
        expect += Jv.T @ F + Jw.T @ N
        power += F @ vc + N @ w
    assert np.allclose(h1 - h0, expect, rtol=1e-12, atol=1e-12) and power > 0
    s0, s1 = s.copy(), s.copy()
    dt = 1e-4
    abd.substep(m, s0, np.zeros(17), abd.Params(dt=dt, self_collision=False, gravity=0.0))
    abd.substep(m, s1, np.zeros(17), abd.Params(dt=dt, self_collision=False, gravity=0.0, body_damping=(0.04, 0.04)))
    dT = abd.energy(m, s1)[0] - abd.energy(m, s0)[0]
    assert dT == pytest.approx(-power * dt, rel=2e-2)
    s2 = s.copy()
    abd.substep(m, s2, np.zeros(17), abd.Params(dt=dt, self_collision=False, gravity=0.0, body_damping=(0.0, 0.0)))
    assert np.array_equal(s2.u(), s0.u())


def test_a1_like_robot_stands_on_its_toes_in_the_numpy_engine():
    """PD-held default pose (0, 0.9, -1.8) x 4 dropped from 2 cm: the four toe proxies carry the robot (feet flags 1, no
    'bad' contact point), per-proxy friction = plane 5 x toe 1 keeps it from sliding, nothing sinks."""
    m = load_urdf(a1_like_urdf(), foot_links=A1_LIKE_TOES, joint_order=MOTOR_NAMES, root_pose=((0, 0, 0.30), None))
    mu = np.where(m.sph_foot >= 0, 1.0, m.sph_friction)
    prm = abd.Params(dt=0.002, substeps=1, iterations=23, friction=5.0, erp=0.2, sphere_friction=mu, self_collision=False, gravity=10.0)
    s = abd.State(m)
    target = np.array([0, 0.9, -1.8] * 4, float)
    s.q = target.copy()
    for k in range(500):
        tau = np.clip(80.0 * (target - s.q) - 1.5 * s.qd, -33.5, 33.5)
        touching = abd.substep(m, s, tau, prm)
    feet = {int(m.sph_foot[g]) for g in touching}
    assert feet == {0, 1, 2, 3}, feet                                    # only toe proxies touch
    assert 0.2 < s.pos[2] < 0.3 and np.abs(s.u()).max() < 0.25 and abs(s.pos[0]) < 0.02
    kin = abd.kinematics(m, s)
    low = min((kin["o"][b] + kin["R"][b] @ m.sph_pos[g])[2] - m.sph_radius[g] for g, b in enumerate(m.sph_body))
    assert low > -2e-3


def test_relative_contact_margins_follow_bullets_rule():
    """mjcf.contact_margins(model, "relative") = Bullet's default contact-breaking threshold per LINK (DESIGN.md §3.4's table):
    gContactBreakingThreshold (0.02) x btCollisionShape::getAngularMotionDisc() of the link's compound shape = half the diagonal of
    the shapes' axis-aligned bounding box + the distance of the box's centre from the shape's origin — by hand for an A1-like
    toe (a 2 cm sphere at the link origin: 0.02 x sqrt(3) x 0.02 = 0.69 mm) and calf (a 0.2 x 0.016 x 0.016 box centred 0.1 below the
    link origin), per ORIGINAL link although the toe is merged into the calf's body; for an MJCF body from its capsules; and the
    absolute forms."""
    from metagym_amd.metalocomotion.mjcf import CONTACT_BREAKING_THRESHOLD, angular_motion_discs, contact_margins
    m = load_urdf(a1_like_urdf(), foot_links=A1_LIKE_TOES, joint_order=MOTOR_NAMES)
    rel = contact_margins(m, "relative")
    assert rel.shape == (len(m.sph_body),) and np.array_equal(rel, CONTACT_BREAKING_THRESHOLD * angular_motion_discs(m))
    toe = np.asarray(m.sph_foot) >= 0
    assert toe.sum() == 4 and np.allclose(rel[toe], 0.02 * np.sqrt(3.0) * 0.02, rtol=1e-12)         # 0.69 mm
    link_names = list(m.link_names)
    calf = np.array([link_names[k] == "FR_lower" for k in m.sph_link])
    assert calf.sum() == 8                                                                           # the box's corners
    # the calf: box 0.2 x 0.016 x 0.016 turned so that its long axis is the link's z, centred at (0, 0, -0.1); inertial origin
    # (0.0065, 0, -0.1073), no rotation: half diagonal |(0.008, 0.008, 0.1)| + |(-0.0065, 0, 0.0073)| = 0.11041 -> 2.21 mm
    assert np.all(np.abs(rel[calf] - rel[calf][0]) < 1e-15)
    by_hand = 0.02 * (np.linalg.norm([0.008, 0.008, 0.1]) + np.linalg.norm([-0.0065, 0.0, 0.0073]))
    assert np.isclose(rel[calf][0], by_hand, rtol=1e-9), (rel[calf][0], by_hand)
    # standing on its toes with the calves at 0.9 - 1.8 rad, a calf corner is ~1.4 cm above the floor: outside the calf's own margin,
    # inside a flat 2 cm one — the reason the flat margin produced 8 "bad" contacts per standing robot (profiles/EXPERIMENTS.md)
    assert rel[calf][0] < 0.014 < 0.02
    # MJCF: the body is the link, geoms in the body frame
    from walker_fixtures import load_models
    h = load_models()["humanoid"]
    d = angular_motion_discs(h)
    lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
    b = int(h.sph_body[-1])
    for gb, p0, p1, r in zip(h.geom_body, h.geom_p0, h.geom_p1, h.geom_radius):
        if int(gb) == b:
            lo, hi = np.minimum(lo, np.minimum(p0, p1) - r), np.maximum(hi, np.maximum(p0, p1) + r)
    assert np.isclose(d[-1], 0.5 * np.linalg.norm(hi - lo) + np.linalg.norm(0.5 * (hi + lo)), rtol=1e-14)
    assert 0.002 < contact_margins(h, "relative").min() and contact_margins(h, "relative").max() < 0.01
    assert np.array_equal(contact_margins(h, 0.0), np.zeros(len(h.sph_body))) and np.array_equal(contact_margins(h, 0.02), np.full(len(h.sph_body), 0.02))
    with pytest.raises(ValueError, match="relative"):
        contact_margins(h, "bullet")
