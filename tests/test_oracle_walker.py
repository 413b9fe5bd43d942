"""oracle/abd.py (numpy restatement of the articulated-body step) — physical invariants.
The physics of this path cannot be pinned against the reference (PyBullet is not in the reference
tree), so the oracle itself is validated by what any correct multibody step must satisfy."""
import copy

import numpy as np
import pytest

from oracle import abd
from walker_fixtures import load_models

# The engine's invariants are checked on MuJoCo's reading of the robot files ("<name>@mujoco"): joints with damping and
# rotor inertia come to rest, which several of the statements below need ("settles", "stays put on the ramp"). The default
# "bullet" reading of the same files (no joint damping) runs through the same code; GPU == oracle is checked on both.
MODELS = {k[:-len("@mujoco")]: v for k, v in load_models().items() if k.endswith("@mujoco")}


def _frictionless_free(m):
    m = copy.deepcopy(m)
    m.joint_damping[:] = 0
    m.joint_stiffness[:] = 0
    m.joint_lo[:] = -100
    m.joint_hi[:] = 100
    return m


def _random_state(m, seed, z=10.0):
    rs = np.random.RandomState(seed)
    s = abd.State(m)
    s.pos[2] = z
    nj = len(m.joint_lo)
    s.q = rs.uniform(-0.3, 0.3, nj)
    s.qd = rs.uniform(-2, 2, nj)
    s.w = rs.uniform(-1, 1, 3)
    s.v = rs.uniform(-1, 1, 3)
    return s


def test_parsed_models_are_sane():
    h = MODELS["humanoid"]
    assert len(h.body_parent) == 13 and len(h.joint_body) == 17 and len(h.foot_body) == 2
    assert h.joint_names[:4] == ["abdomen_z", "abdomen_y", "abdomen_x", "right_hip_x"]     # humanoids.py:18-28 order
    assert 35 < h.body_mass.sum() < 55                   # ~42 kg of density-1000 capsules
    assert abs(h.body_pos[0][2] - 1.4) < 1e-12           # humanoid.xml:16
    a = MODELS["ant"]
    assert len(a.joint_body) == 8 and len(a.foot_body) == 4 and abs(a.geom_friction - 1.5) < 1e-12
    for m in MODELS.values():
        for I in m.body_inertia:                          # symmetric positive definite
            assert np.allclose(I, I.T) and np.all(np.linalg.eigvalsh(I) > 0)
        assert np.allclose(np.linalg.norm(m.joint_axis, axis=1), 1.0)
        assert np.all(m.joint_lo < m.joint_hi)


@pytest.mark.parametrize("name", ["humanoid", "ant"])
def test_mass_matrix_matches_kinetic_energy(name):
    """0.5 u^T M u == sum_b 0.5 m |v_c|^2 + 0.5 w^T I w, and M is symmetric positive definite."""
    m = _frictionless_free(MODELS[name])
    s = _random_state(m, 1)
    M, h, kin, _ = abd.mass_matrix_and_bias(m, s)
    u = s.u()
    ke = 0.0
    for b in range(len(m.body_parent)):
        vc = abd.point_jacobian(m, kin, b, kin["c"][b]) @ u
        wb = abd.angular_jacobian(m, kin, b) @ u
        Iw = kin["R"][b] @ m.body_inertia[b] @ kin["R"][b].T
        ke += 0.5 * m.body_mass[b] * vc @ vc + 0.5 * wb @ Iw @ wb
    ke += 0.5 * np.sum(m.joint_armature * s.qd ** 2)
    assert abs(0.5 * u @ M @ u - ke) < 1e-9 * ke
    assert np.allclose(M, M.T) and np.all(np.linalg.eigvalsh(M) > 0)


@pytest.mark.parametrize("name", ["humanoid", "ant"])
def test_free_flight_conserves_momentum_and_energy(name):
    """No gravity, no damping, no limits: linear & angular momentum and energy are conserved; the
    first-order integrator's drift halves when dt halves."""
    m = _frictionless_free(MODELS[name])
    g = abd.GRAVITY.copy()
    abd.GRAVITY[:] = 0
    try:
        drifts = []
        for dt in (0.001, 0.0005):
            s = _random_state(m, 0)
            T0, _ = abd.energy(m, s)
            P0, L0 = abd.momentum(m, s)
            prm = abd.Params(dt=dt, self_collision=False)
            for _ in range(int(round(0.05 / dt))):
                abd.substep(m, s, np.zeros(len(m.joint_lo)), prm)
            T1, _ = abd.energy(m, s)
            P1, L1 = abd.momentum(m, s)
            drifts.append((abs(T1 - T0) / T0, np.abs(P1 - P0).max() / np.abs(P0).max(),
                           np.abs(L1 - L0).max() / np.abs(L0).max()))
        for a, b in zip(*drifts):
            assert a < 2e-3 and b < 0.6 * a + 1e-9, drifts
    finally:
        abd.GRAVITY[:] = g


def test_free_fall_gains_momentum_mg_t():
    m = _frictionless_free(MODELS["humanoid"])
    s = _random_state(m, 3)
    P0, _ = abd.momentum(m, s)
    prm = abd.Params(dt=0.001)
    for _ in range(100):
        abd.substep(m, s, np.zeros(17), prm)
    P1, _ = abd.momentum(m, s)
    expect = -9.8 * m.body_mass.sum() * 0.1
    assert abs((P1 - P0)[2] - expect) < 0.02 * abs(expect)
    assert np.abs((P1 - P0)[:2]).max() < 0.02 * abs(expect)


def test_instantaneous_momentum_balance():
    """With du/dt = -M^-1 h (no torques) the total momentum changes at exactly m*g: checks the bias term."""
    m = _frictionless_free(MODELS["humanoid"])
    s = _random_state(m, 5)
    M, h, _, _ = abd.mass_matrix_and_bias(m, s)
    ud = -np.linalg.solve(M, h)
    eps = 1e-6
    s2 = s.copy()
    u = s.u() + eps * ud
    s2.v, s2.w, s2.qd = u[:3], u[3:6], u[6:]
    abd.integrate_positions(s2, eps)
    P0, L0 = abd.momentum(m, s)
    P1, L1 = abd.momentum(m, s2)
    dP = (P1 - P0) / eps
    assert np.allclose(dP, [0, 0, -9.8 * m.body_mass.sum()], atol=2e-3 * 9.8 * m.body_mass.sum())


@pytest.mark.parametrize("name,feet_z", [("humanoid", 0.0), ("ant", 0.0)])
def test_drop_on_ground_settles_without_sinking(name, feet_z):
    """Zero action from the reset pose: the robot falls / collapses onto the plane; no collision
    sphere ends more than 5 mm below the ground, velocities stay bounded, and once at rest the
    contact impulses carry the weight (vertical momentum stops changing)."""
    m = MODELS[name]
    env = abd.WalkerEnv(m, motor_power=np.full(len(m.joint_lo), 100.0) if name == "ant" else abd.HUMANOID_MOTOR_POWER,
                        alive_z=-1.0)
    env.reset(np.zeros(len(m.joint_lo)))
    worst_pen, vmax = 0.0, 0.0
    for t in range(150):
        env.step(np.zeros(len(m.joint_lo)))
        kin = abd.kinematics(m, env.s)
        for g, b in enumerate(m.sph_body):
            x = kin["o"][b] + kin["R"][b] @ m.sph_pos[g]
            worst_pen = max(worst_pen, m.sph_radius[g] - x[2])
        vmax = max(vmax, np.abs(env.s.u()).max())
    assert worst_pen < 0.02, worst_pen
    assert vmax < 80.0, vmax
    P, _ = abd.momentum(m, env.s)
    assert abs(P[2]) < 0.05 * 9.8 * m.body_mass.sum() * 0.02 * 50      # essentially at rest vertically


def test_segment_closest_points():
    rs = np.random.RandomState(0)
    for k in range(300):
        p1, q1, p2, q2 = rs.uniform(-1, 1, (4, 3))
        if k % 5 == 0:
            q1 = p1.copy()                       # sphere vs capsule
        if k % 7 == 0:
            q2 = p2.copy()
        ca, cb = abd.segment_closest(p1, q1, p2, q2)
        ss, tt = np.meshgrid(np.linspace(0, 1, 33), np.linspace(0, 1, 33))
        brute = np.linalg.norm((p1 + ss[..., None] * (q1 - p1)) - (p2 + tt[..., None] * (q2 - p2)), axis=-1).min()
        assert np.linalg.norm(ca - cb) <= brute + 1e-9


def test_self_collision_is_an_internal_force():
    """Swing both thighs inwards in free space until they touch: every self-contact row (normal and
    friction) must be momentum-neutral — A M^-1 J^T = 0 with A the robot's momentum map — the contact
    must stop the inter-penetration, and the total momentum must stay where it was."""
    m = _frictionless_free(MODELS["humanoid"])
    m.joint_armature[:] = 0      # rotor inertia (MJCF armature) is not rigid-body momentum; pure rigid bodies here
    g = abd.GRAVITY.copy()
    abd.GRAVITY[:] = 0
    try:
        prm = abd.Params(dt=0.001)
        s = abd.State(m)
        s.pos[2] = 10.0
        s.qd[m.joint_names.index("right_hip_x")] = 2.0
        s.qd[m.joint_names.index("left_hip_x")] = 2.0
        P0, L0 = abd.momentum(m, s)
        seen, max_depth = False, 0.0
        for _ in range(120):
            kin = abd.kinematics(m, s)
            rows = [r for r in abd.constraint_rows(m, s, kin, prm) if r[4] == -2]
            if rows and not seen:
                seen = True
                M, _, _, _ = abd.mass_matrix_and_bias(m, s, kin)
                A = np.zeros((6, M.shape[0]))
                for b in range(len(m.body_parent)):
                    Jv = abd.point_jacobian(m, kin, b, kin["c"][b])
                    Jw = abd.angular_jacobian(m, kin, b)
                    Iw = kin["R"][b] @ m.body_inertia[b] @ kin["R"][b].T
                    A[:3] += m.body_mass[b] * Jv
                    A[3:] += abd.skew(kin["c"][b]) @ (m.body_mass[b] * Jv) + Iw @ Jw
                for r in rows:
                    assert np.abs(A @ np.linalg.solve(M, r[0])).max() < 1e-12
            if rows:
                max_depth = max(max_depth, max(r[1] for r in rows if r[2] == 0) * prm.dt / prm.erp)
            abd.substep(m, s, np.zeros(17), prm)
        assert seen and max_depth < 2e-3, (seen, max_depth)
        P1, L1 = abd.momentum(m, s)
        # the impulses themselves are exactly neutral (asserted above); what remains is the integrator's
        # one-step O(dt * centrifugal force) term that is no longer cancelled by the position update when
        # an inelastic impact removes the velocity in between (~30 N * 1 ms here)
        assert np.abs(P1 - P0).max() < 0.05 and np.abs(L1 - L0).max() < 0.05
    finally:
        abd.GRAVITY[:] = g


def _box(pos, half, mu=1.0, pitch=0.0):
    c, s = np.cos(pitch), np.sin(pitch)
    R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])        # rotation about y by `pitch`
    return (np.array(pos, float), R, np.array(half, float), mu)


def test_sphere_box_contact_cases():
    box = _box([1.0, 0.0, 0.5], [0.5, 2.0, 0.1])
    depth, n, xc = abd.sphere_box(np.array([1.2, 0.3, 0.65]), 0.1, box)            # above the top face
    assert np.isclose(depth, 0.05) and np.allclose(n, [0, 0, 1]) and np.allclose(xc, [1.2, 0.3, 0.6])
    depth, n, xc = abd.sphere_box(np.array([1.56, 0.0, 0.66]), 0.1, box)           # over the +x edge: normal along the diagonal
    assert np.allclose(n, np.array([0.06, 0, 0.06]) / np.hypot(0.06, 0.06)) and np.isclose(depth, 0.1 - np.hypot(0.06, 0.06))
    depth, n, xc = abd.sphere_box(np.array([1.0, 0.0, 0.58]), 0.05, box)           # centre inside: out through the nearest face
    assert np.isclose(depth, 0.05 + 0.02) and np.allclose(n, [0, 0, 1]) and np.allclose(xc, [1.0, 0.0, 0.6])
    assert abd.sphere_box(np.array([3.0, 0.0, 0.5]), 0.1, box)[0] < 0                # far away
    tilted = _box([0, 0, 0], [5, 5, 0.01], pitch=-0.3)                              # a ramp rising towards +x
    depth, n, xc = abd.sphere_box(np.array([1.0, 0.0, 1.0 * np.tan(0.3) + 0.05]), 0.1, tilted)
    assert np.allclose(n, [-np.sin(0.3), 0, np.cos(0.3)]) and depth > 0


@pytest.mark.parametrize("name", ["ant"])
def test_drop_on_a_box_rests_on_its_top_face(name):
    """A platform 0.4 m above the ground plane: the robot dropped over it comes to rest on the platform (no sphere deeper than
    2 cm in the top face, none anywhere near the ground plane), and the contact impulses carry its weight."""
    m = MODELS[name]
    top = 0.4
    prm = abd.Params(terrain=[_box([0.0, 0.0, top - 0.05], [2.0, 2.0, 0.05], mu=0.8)])
    env = abd.WalkerEnv(m, prm=prm, motor_power=np.full(len(m.joint_lo), 100.0), alive_z=-1.0)
    env.reset(np.zeros(len(m.joint_lo)))
    env.s.pos[2] += top
    lowest = np.inf
    for t in range(150):
        env.step(np.zeros(len(m.joint_lo)))
        kin = abd.kinematics(m, env.s)
        lowest = min(lowest, min((kin["o"][b] + kin["R"][b] @ m.sph_pos[g])[2] - m.sph_radius[g] for g, b in enumerate(m.sph_body)))
    assert top - 0.02 < lowest
    P, _ = abd.momentum(m, env.s)
    assert abs(P[2]) < 0.05 * 9.8 * m.body_mass.sum()


@pytest.mark.parametrize("mu,slides", [(1.0, False), (0.1, True)])
def test_friction_cone_on_a_ramp(mu, slides):
    """A 0.3 rad ramp (tan = 0.31) with the box's own friction coefficient: at mu = 1.0 the dropped robot comes to rest and
    stays; at mu = 0.1 it slides downhill with the acceleration of a block on an incline, g (sin - mu cos)."""
    m = MODELS["ant"]
    ang = 0.3
    ramp = _box([0.0, 0.0, 0.0], [30.0, 5.0, 0.02], mu=mu, pitch=-ang)             # rises towards +x through the origin
    env = abd.WalkerEnv(m, prm=abd.Params(terrain=[ramp]), motor_power=np.full(len(m.joint_lo), 100.0), alive_z=-1.0)
    env.reset(np.zeros(len(m.joint_lo)))
    env.s.pos[:] = [5.0, 0.0, 5.0 * np.tan(ang) + env.s.pos[2] + 0.45]             # feet 5 cm above the ramp
    zero = np.zeros(len(m.joint_lo))
    if not slides:
        for t in range(110):
            env.step(zero)
        x0 = env.s.pos[0]
        for t in range(25):
            env.step(zero)
        assert abs(env.s.pos[0] - x0) < 0.03 and np.abs(env.s.u()).max() < 0.3
    else:
        for t in range(30):
            env.step(zero)
        v0 = env.s.v.copy()
        for t in range(50):      # 1 s on the ramp
            env.step(zero)
        along = (env.s.v - v0) @ np.array([np.cos(ang), 0.0, np.sin(ang)])
        assert along == pytest.approx(-9.8 * (np.sin(ang) - mu * np.cos(ang)), rel=0.25)
