"""oracle/abd.py — numpy (float64) restatement of the articulated-body step used for MetaLocomotion.

TEST INFRASTRUCTURE (see oracle/__init__.py). **Parity unpinned**: in the reference the arithmetic of
this path is `pybullet.stepSimulation()` (metalocomotion/envs/utils/scene_bases.py:50;
pybullet>=3.0.7, setup.py:57,59), a third-party C++ library that is neither vendored in the
reference nor installable here, and the reference's tests at that boundary assert nothing
(metalocomotion/test.py). What is restated is the *published algorithm family* Bullet's
btMultiBodyDynamicsWorld implements, with the reference's own parameters:

  * reduced-coordinate multibody dynamics, floating base + hinge joints (Featherstone, "Rigid Body
    Dynamics Algorithms", 2008): joint-space inertia matrix M(q) and bias forces h(q, u) — built here
    from world-frame body Jacobians and a Newton-Euler pass instead of ABA (same equations);
  * semi-implicit Euler at 4 sub-steps of 5 ms per env step (walker_base_env.py:7, scene_bases.py:56);
  * velocity-level contact / joint-limit constraints solved by projected Gauss-Seidel ("sequential
    impulses", Catto 2005) on the Delassus matrix J M^-1 J^T, 5 iterations (scene_bases.py:17),
    Baumgarte position correction with ERP 0.9 (scene_bases.py:55), gravity 9.8 (env_bases.py:48),
    ground = plane z=0 with lateral friction 0.8 (stadium.py:23) combined multiplicatively with the
    geoms' 0.8 (humanoid.xml:5) like Bullet does.

Stated assumptions (Bullet internals that cannot be checked here): MJCF joint `damping`,
`stiffness`, `armature` act as in MuJoCo's documentation (explicit damping/spring torque, armature
added to the diagonal of M); link inertia from solid-capsule / sphere formulas at density 1000;
TORQUE_CONTROL torques persist across the 4 internal sub-steps; self-collision (robot_bases.py:119
flags) as capsule-capsule contacts between bodies that are neither ancestor-related nor welded.

The Python-side rules around the physics ARE pinned by the reference source and are restated
exactly: torque = power * 0.41 * clip(a) (humanoids.py:50-54), observation (walker_base.py:31-64),
reward / done (walker_base_env.py:43-82), reset noise U(-0.1, 0.1) on every joint (walker_base.py:15).

Pure numpy with python loops: fine for the handful of envs the tests push through it.
"""
import numpy as np

GRAVITY = np.array([0.0, 0.0, -9.8])


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def rodrigues(axis, angle):
    k = skew(axis)
    return np.eye(3) + np.sin(angle) * k + (1.0 - np.cos(angle)) * (k @ k)


class State(object):
    def __init__(self, model):
        nj = len(model.joint_lo)
        self.pos = np.array(model.body_pos[0], float)     # base body origin, world
        self.rot = np.array(model.body_rot[0], float)     # base body orientation, world
        self.v = np.zeros(3)                               # base origin velocity, world
        self.w = np.zeros(3)                               # base angular velocity, world
        self.q = np.zeros(nj)
        self.qd = np.zeros(nj)

    def copy(self):
        s = State.__new__(State)
        for k in ("pos", "rot", "v", "w", "q", "qd"):
            setattr(s, k, getattr(self, k).copy())
        return s

    def u(self):
        return np.concatenate([self.v, self.w, self.qd])


def kinematics(m, s):
    """World frames. Returns dict: R[b], o[b], c[b] (com), p[j] (anchor), a[j] (axis)."""
    nb, nj = len(m.body_parent), len(m.joint_body)
    R, o = [None] * nb, [None] * nb
    p, a = np.zeros((nj, 3)), np.zeros((nj, 3))
    for b in range(nb):
        if m.body_parent[b] < 0:
            Rc, oc = s.rot.copy(), s.pos.copy()
        else:
            pb = m.body_parent[b]
            Rc = R[pb] @ m.body_rot[b]
            oc = o[pb] + R[pb] @ m.body_pos[b]
        for j in np.nonzero(m.joint_body == b)[0]:
            p[j] = oc + Rc @ m.joint_anchor[j]
            a[j] = Rc @ m.joint_axis[j]
            Rn = Rc @ rodrigues(m.joint_axis[j], s.q[j])
            oc = p[j] - Rn @ m.joint_anchor[j]
            Rc = Rn
        R[b], o[b] = Rc, oc
    c = np.array([o[b] + R[b] @ m.body_com[b] for b in range(nb)])
    return dict(R=np.array(R), o=np.array(o), c=c, p=p, a=a)


def ancestors_joints(m, b):
    """indices of every joint between the base and body b (inclusive of b's own joints)."""
    out = []
    while b >= 0:
        out.extend(np.nonzero(m.joint_body == b)[0].tolist())
        b = m.body_parent[b]
    return sorted(out)


def point_jacobian(m, kin, b, x):
    """3 x n Jacobian of the world velocity of point x attached to body b w.r.t. u = [v0, w0, qd]."""
    n = 6 + len(m.joint_body)
    J = np.zeros((3, n))
    J[:, 0:3] = np.eye(3)
    J[:, 3:6] = -skew(x - kin["o"][0])
    for j in ancestors_joints(m, b):
        J[:, 6 + j] = np.cross(kin["a"][j], x - kin["p"][j])
    return J


def angular_jacobian(m, kin, b):
    n = 6 + len(m.joint_body)
    J = np.zeros((3, n))
    J[:, 3:6] = np.eye(3)
    for j in ancestors_joints(m, b):
        J[:, 6 + j] = kin["a"][j]
    return J


def mass_matrix_and_bias(m, s, kin=None, gravity=None, body_damping=(0.0, 0.0)):
    """M(q) (with armature) and h(q,u) such that M du/dt + h = tau_generalised (gravity inside h).
    body_damping = (k_lin, k_ang): btMultiBody's velocity damping of every body — force -m v (k + k |v|) at the centre of
    mass, torque -I w (k + k |w|) (btMultiBody::computeAccelerationsArticulatedBodyAlgorithmMultiDof adds exactly this to
    every link's zero-acceleration force, DAMPING_K1 = DAMPING_K2 = m_linearDamping / m_angularDamping, default 0.04)."""
    kin = kin or kinematics(m, s)
    # gravity: None = the module's (0, 0, -9.8); a number g = (0, 0, -g); a 3-vector = the acceleration itself (setGravity(gx, gy, gz))
    grav = GRAVITY if gravity is None else (np.asarray(gravity, float) if np.ndim(gravity) == 1 else np.array([0.0, 0.0, -float(gravity)]))
    k_lin, k_ang = body_damping
    u_all = s.u()
    nb, nj = len(m.body_parent), len(m.joint_body)
    n = 6 + nj
    M = np.zeros((n, n))
    h = np.zeros(n)
    # velocity-product accelerations (du/dt = 0), body by body
    frame = [None] * nb       # (w, alpha, x_ref, a_ref) after the body's own joints
    for b in range(nb):
        if m.body_parent[b] < 0:
            w, al, xr, ar = s.w.copy(), np.zeros(3), kin["o"][0].copy(), np.zeros(3)
        else:
            w, al, xr, ar = [x.copy() for x in frame[m.body_parent[b]]]
        for j in np.nonzero(m.joint_body == b)[0]:
            r = kin["p"][j] - xr
            ar = ar + np.cross(al, r) + np.cross(w, np.cross(w, r))
            xr = kin["p"][j].copy()
            wj = kin["a"][j] * s.qd[j]
            al = al + np.cross(w, wj)
            w = w + wj
        frame[b] = (w, al, xr, ar)
        r = kin["c"][b] - xr
        a_c = ar + np.cross(al, r) + np.cross(w, np.cross(w, r))
        Iw = kin["R"][b] @ m.body_inertia[b] @ kin["R"][b].T
        Jv = point_jacobian(m, kin, b, kin["c"][b])
        Jw = angular_jacobian(m, kin, b)
        M += m.body_mass[b] * Jv.T @ Jv + Jw.T @ Iw @ Jw
        h += Jv.T @ (m.body_mass[b] * (a_c - grav)) + Jw.T @ (Iw @ al + np.cross(w, Iw @ w))
        if k_lin != 0.0 or k_ang != 0.0:
            vc = Jv @ u_all
            h += Jv.T @ (m.body_mass[b] * (k_lin + k_lin * np.linalg.norm(vc)) * vc) \
                + Jw.T @ ((k_ang + k_ang * np.linalg.norm(w)) * (Iw @ w))
    M[np.arange(6, n), np.arange(6, n)] += m.joint_armature
    return M, h, kin, frame


def integrate_positions(s, dt):
    s.pos = s.pos + dt * s.v
    ang = np.linalg.norm(s.w) * dt
    if ang > 0:
        s.rot = rodrigues(s.w / np.linalg.norm(s.w), ang) @ s.rot
    s.q = s.q + dt * s.qd


class Params(object):
    def __init__(self, dt=0.005, substeps=4, iterations=5, erp=0.9, friction=0.8 * 0.8, power=0.41, max_contacts=12,
                 limit_erp=0.2, self_collision=True, self_friction=0.8 * 0.8, terrain=(), gravity=None, sphere_friction=None,
                 body_damping=(0.0, 0.0), max_velocity=0.0, contact_margin=0.0):
        # Bullet's contact-breaking margin: a proxy within this distance ABOVE the ground or a terrain box is a contact point — a
        # speculative solver row (contact_bias) and a feet_contact flag. 0 = penetration only
        #   (a scalar: that margin for every proxy; an array [n proxies]: per proxy, e.g. mjcf.contact_margins(m, "relative") —
        #   Bullet's default is relative, 0.02 x the link's angular motion disc)
        self.contact_margin = float(contact_margin) if np.ndim(contact_margin) == 0 else np.asarray(contact_margin, float)
        self.max_velocity = float(max_velocity)   # btMultiBody's m_maxCoordinateVelocity (100 in Bullet): clamp of every generalized
        #                                           velocity at the end of a sub-step (mg_walker_params.max_coordinate_velocity); 0 = off
        self.terrain = list(terrain)          # static boxes on top of the ground plane: (position[3], R[3,3] box->world, half_extents[3], mu)
        self.gravity = None if gravity is None else (np.asarray(gravity, float) if np.ndim(gravity) == 1 else float(gravity))   # None: the module's GRAVITY (9.8, env_bases.py:48); a 3-vector: the acceleration itself
        # per-proxy lateral friction (mg_walker_params.sphere_friction): `friction` is then the ground's own coefficient and
        # every terrain box's mu its own; the contact's coefficient is the product with the proxy's link (Bullet multiplies)
        self.sphere_friction = None if sphere_friction is None else np.asarray(sphere_friction, float)
        self.body_damping = (float(body_damping[0]), float(body_damping[1]))
        self.dt, self.substeps, self.iterations, self.erp, self.friction, self.power = dt, substeps, iterations, erp, friction, power
        self.limit_erp = limit_erp            # Bullet's default constraint ERP (btContactSolverInfo::m_erp2 = 0.2);
        #                                       setDefaultContactERP(0.9) only changes the contact ERP
        self.self_collision, self.self_friction = self_collision, self_friction   # geom friction squared
        self.max_contacts = max_contacts      # engine limit: of the ground, terrain and self-contact candidates the max_contacts DEEPEST are kept (select_contacts)


def segment_closest(p1, q1, p2, q2):
    """Closest points of segments [p1,q1] and [p2,q2] (Ericson, Real-Time Collision Detection 5.1.9);
    degenerate segments (spheres) included."""
    d1, d2, r = q1 - p1, q2 - p2, p1 - p2
    a, e, f = d1 @ d1, d2 @ d2, d2 @ r
    eps = 1e-12
    if a <= eps and e <= eps:
        return p1, p2
    if a <= eps:
        sc, tc = 0.0, min(max(f / e, 0.0), 1.0)
    else:
        c = d1 @ r
        if e <= eps:
            tc, sc = 0.0, min(max(-c / a, 0.0), 1.0)
        else:
            b = d1 @ d2
            den = a * e - b * b
            sc = min(max((b * f - c * e) / den, 0.0), 1.0) if den > eps else 0.0
            tc = (b * sc + f) / e
            if tc < 0.0:
                tc, sc = 0.0, min(max(-c / a, 0.0), 1.0)
            elif tc > 1.0:
                tc, sc = 1.0, min(max((b - c) / a, 0.0), 1.0)
    return p1 + sc * d1, p2 + tc * d2


def sphere_box(x, radius, box):
    """Deepest-point contact of a sphere (centre x) with a static oriented box (position, R, half extents, mu):
    returns (depth, normal, point on the box surface) — depth <= 0: no contact. Outside the box the normal points from the
    closest surface point to the centre; with the centre inside, along the face of least penetration (first of x, y, z on ties)."""
    p, R, h, _ = box
    l = R.T @ (x - p)
    c = np.minimum(np.maximum(l, -h), h)
    d = l - c
    dist2 = d @ d
    if dist2 > 0.0:
        dist = np.sqrt(dist2)
        return radius - dist, R @ (d / dist), p + R @ c
    k = int(np.argmin(h - np.abs(l)))
    sgn = 1.0 if l[k] >= 0.0 else -1.0
    n_l = np.zeros(3)
    n_l[k] = sgn
    c = l.copy()
    c[k] = sgn * h[k]
    return radius + (h[k] - abs(l[k])), R @ n_l, p + R @ c


def tangent_basis(nrm):
    """Two unit tangents orthogonal to nrm (deterministic choice shared with the kernels)."""
    ref = np.array([1.0, 0.0, 0.0]) if abs(nrm[0]) < 0.9 else np.array([0.0, 1.0, 0.0])
    t1 = np.cross(nrm, ref)
    t1 = t1 / np.linalg.norm(t1)
    return t1, np.cross(nrm, t1)


MAX_CANDIDATES = 48     # contact candidates collected per sub-step before the selection (walker.hip W_MAXCAND): later ones are dropped


def contact_bias(depth, prm):
    """Target velocity along a contact normal (the row is J u >= bias). Penetration (depth >= 0): Baumgarte push-out
    erp * depth / dt. Separation inside the contact margin (depth < 0): the SPECULATIVE row of Bullet's
    setupMultiBodyContactConstraint / setupContactConstraint — `penetration = distance + slop > 0`: positional error 0 and
    velocityError -= penetration / dt — i.e. the proxy may approach the surface by at most its gap per step: depth / dt (< 0),
    no ERP."""
    return (prm.erp * depth if depth >= 0.0 else depth) / prm.dt


def contact_candidates(m, s, kin, prm):
    """Contact candidates of one configuration in CANDIDATE ORDER — ground plane per collision proxy (proxy order), then terrain
    (per proxy the deepest box, first on ties), then self-collision pairs (pair order) — as dicts, and the set of proxies
    `touching` = every proxy with a ground / terrain candidate: what pybullet.getContactPoints reports (walker_base_env.py:57-63
    via robot_bases.py:291-292: all manifold points within the contact-breaking margin), whether or not the solver's contact
    cap kept the point.

    prm.contact_margin (mg_walker_params.contact_margin / sphere_margin; 0 = penetration only; Bullet: 0.02 x the link's angular
    motion disc, metalocomotion.mjcf.contact_margins): a proxy is a ground / terrain candidate while depth > -its margin.
    Self-collision pairs stay penetration-only."""
    mg_all = getattr(prm, "contact_margin", 0.0)
    margin_of = (lambda g: float(mg_all)) if np.ndim(mg_all) == 0 else (lambda g: float(mg_all[g]))
    cands, touching = [], set()
    for g in range(len(m.sph_body)):
        b = m.sph_body[g]
        x = kin["o"][b] + kin["R"][b] @ m.sph_pos[g]
        depth = m.sph_radius[g] - x[2]
        if depth > -margin_of(g):
            touching.add(g)
            cands.append(dict(cat=0, g=g, depth=depth, xc=np.array([x[0], x[1], 0.0]), nrm=np.array([0.0, 0.0, 1.0])))
    if prm.terrain:
        for g in range(len(m.sph_body)):
            b = m.sph_body[g]
            x = kin["o"][b] + kin["R"][b] @ m.sph_pos[g]
            best = None
            for box in prm.terrain:
                depth, nrm, xc = sphere_box(x, m.sph_radius[g], box)
                if depth > -margin_of(g) and (best is None or depth > best[0]):
                    best = (depth, nrm, xc, box[3])
            if best is not None:
                touching.add(g)
                cands.append(dict(cat=1, g=g, depth=best[0], nrm=best[1], xc=best[2], mu=best[3]))
    if prm.self_collision and hasattr(m, "pair_a"):
        for ga, gb in zip(m.pair_a, m.pair_b):
            ba, bb = m.geom_body[ga], m.geom_body[gb]
            a0 = kin["o"][ba] + kin["R"][ba] @ m.geom_p0[ga]
            a1 = kin["o"][ba] + kin["R"][ba] @ m.geom_p1[ga]
            b0 = kin["o"][bb] + kin["R"][bb] @ m.geom_p0[gb]
            b1 = kin["o"][bb] + kin["R"][bb] @ m.geom_p1[gb]
            ca, cb = segment_closest(a0, a1, b0, b1)
            dvec = ca - cb
            dist = np.linalg.norm(dvec)
            depth = m.geom_radius[ga] + m.geom_radius[gb] - dist
            if depth > 0.0 and dist > 1e-9:
                nrm = dvec / dist                                   # from b towards a
                # one contact point for both bodies (middle of the overlap): equal and opposite forces at
                # the same point change neither the linear nor the angular momentum of the robot
                xc = 0.5 * ((ca - m.geom_radius[ga] * nrm) + (cb + m.geom_radius[gb] * nrm))
                cands.append(dict(cat=2, ba=ba, bb=bb, depth=depth, nrm=nrm, xc=xc))
    return cands[:MAX_CANDIDATES], touching


def depth_key(depth):
    """The ranking key of the contact cap: the depth on a 2^-20 m grid (exact in float64: a power-of-two scaling and a floor)."""
    return float(np.floor(depth * 1048576.0))


def select_contacts(cands, max_contacts):
    """The solver's contact cap: with more than `max_contacts` candidates the DEEPEST are kept (ties on depth_key's grid: the earlier candidate),
    and the kept ones stay in candidate order — penetrating points of any kind go before speculative ones."""
    if len(cands) <= max_contacts:
        return list(cands)
    # depths are compared on a 2^-20 m (0.95 um) grid: contact points that are equally deep by symmetry (a robot lying flat, two
    # feet side by side) differ by round-off from one implementation to the next, and must not be ordered by that round-off
    order = sorted(range(len(cands)), key=lambda i: (-depth_key(cands[i]["depth"]), i))
    keep = sorted(order[:max_contacts])
    return [cands[i] for i in keep]


def constraint_rows(m, s, kin, prm, touching_out=None):
    """Rows (J, bias, kind, partner, proxy, normal | mu): kind 0 = unilateral (lambda >= 0), 1/2 = ground friction rows whose
    bound is friction * lambda[partner], -1 = friction row carrying its own coefficient, 3 = self-contact friction. bias is the
    target velocity along the row. `touching_out` (a set) receives the proxies with a ground / terrain candidate."""
    rows = []
    n = 6 + len(m.joint_body)
    cands, touching = contact_candidates(m, s, kin, prm)
    if touching_out is not None:
        touching_out |= touching
    for c in select_contacts(cands, prm.max_contacts):
        k = len(rows)
        if c["cat"] == 0:
            g = c["g"]
            Jc = point_jacobian(m, kin, m.sph_body[g], c["xc"])
            rows.append((Jc[2], contact_bias(c["depth"], prm), 0, -1, g, c["nrm"]))     # (.., proxy, contact normal)
            if prm.sphere_friction is None:
                rows.append((Jc[0], 0.0, 1, k, g))
                rows.append((Jc[1], 0.0, 2, k, g))
            else:
                mu = prm.friction * prm.sphere_friction[g]
                rows.append((Jc[0], 0.0, -1, k, g, mu))
                rows.append((Jc[1], 0.0, -1, k, g, mu))
        elif c["cat"] == 1:      # terrain: friction rows carry the box's own coefficient
            g, nrm, mu = c["g"], c["nrm"], c["mu"]
            if prm.sphere_friction is not None:
                mu = mu * prm.sphere_friction[g]
            Jc = point_jacobian(m, kin, m.sph_body[g], c["xc"])
            t1, t2 = tangent_basis(nrm)
            rows.append((nrm @ Jc, contact_bias(c["depth"], prm), 0, -1, g, nrm))
            rows.append((t1 @ Jc, 0.0, -1, k, g, mu))
            rows.append((t2 @ Jc, 0.0, -1, k, g, mu))
        else:
            # self-collision between the capsule geoms of bodies that are neither ancestor-related nor welded
            # (PyBullet flags at robot_bases.py:119); friction = geom friction squared (Bullet multiplies)
            nrm = c["nrm"]
            Jd = point_jacobian(m, kin, c["ba"], c["xc"]) - point_jacobian(m, kin, c["bb"], c["xc"])
            t1, t2 = tangent_basis(nrm)
            rows.append((nrm @ Jd, contact_bias(c["depth"], prm), 0, -1, -2))
            rows.append((t1 @ Jd, 0.0, 3, k, -2))
            rows.append((t2 @ Jd, 0.0, 3, k, -2))
    for j in range(len(m.joint_body)):
        e = np.zeros(n)
        if s.q[j] < m.joint_lo[j]:
            e[6 + j] = 1.0
            rows.append((e, prm.limit_erp * (m.joint_lo[j] - s.q[j]) / prm.dt, 0, -1, -1))
        elif s.q[j] > m.joint_hi[j]:
            e[6 + j] = -1.0
            rows.append((e, prm.limit_erp * (s.q[j] - m.joint_hi[j]) / prm.dt, 0, -1, -1))
    return rows


def pgs(A, rhs, rows, friction, iterations):
    """Projected Gauss-Seidel on  w = A lam + rhs,  0 <= lam  _|_  w >= 0  (unilateral rows) and
    |lam_t| <= mu * lam_n (friction rows; mu = friction[0] for ground rows (kind 1/2), friction[1] for
    self-contact rows (kind 3)), natural row order, zero warm start."""
    lam = np.zeros(len(rows))
    for _ in range(iterations):
        for r, row in enumerate(rows):
            kind, partner = row[2], row[3]
            if A[r, r] <= 0:
                continue
            x = lam[r] - (A[r] @ lam + rhs[r]) / A[r, r]
            if kind == 0:
                lam[r] = max(0.0, x)
            else:
                lim = (row[5] if kind < 0 else friction[1] if kind == 3 else friction[0]) * lam[partner]
                lam[r] = min(lim, max(-lim, x))
    return lam


def foot_forces(m, rows, lam, dt):
    """|sum over a foot's contact points of normal impulse x normal| / dt, per foot (newtons): what a1.py:325-356
    GetFootContactsForce adds up from PyBullet's contact points (contact[9] normalForce x contact[7] normal)."""
    sph_foot = np.asarray(m.sph_foot)
    f = np.zeros((len(m.foot_body), 3))
    for r, row in enumerate(rows):
        if row[2] == 0 and row[4] >= 0 and sph_foot[row[4]] >= 0:
            f[sph_foot[row[4]]] += lam[r] * row[5]
    return np.linalg.norm(f, axis=1) / dt


def substep(m, s, tau_motor, prm, out=None, ext=None):
    """One 5 ms sub-step. Returns the set of sphere indices in contact. `out` (a dict) receives the constraint rows and their
    multipliers. `ext` = (force[3], point[3]) in the base body frame: an external push on the base during this sub-step
    (pybullet.applyExternalForce(..., LINK_FRAME); mg_walker_params.ext_wrench)."""
    M, h, kin, _ = mass_matrix_and_bias(m, s, gravity=prm.gravity, body_damping=prm.body_damping)
    n = M.shape[0]
    tau = np.zeros(n)
    tau[6:] = tau_motor - m.joint_damping * s.qd - m.joint_stiffness * s.q
    if ext is not None:
        F, r = s.rot @ np.asarray(ext[0], float), s.rot @ np.asarray(ext[1], float)
        tau[0:3] += F
        tau[3:6] += np.cross(r, F)
    u = s.u()
    L = np.linalg.cholesky(M)
    solve = lambda rhs: np.linalg.solve(L.T, np.linalg.solve(L, rhs))
    u_star = u + prm.dt * solve(tau - h)
    touching = set()
    rows = constraint_rows(m, s, kin, prm, touching_out=touching)
    if rows:
        J = np.array([r[0] for r in rows])
        MinvJT = solve(J.T)
        A = J @ MinvJT
        rhs = J @ u_star - np.array([r[1] for r in rows])
        lam = pgs(A, rhs, rows, (prm.friction, prm.self_friction), prm.iterations)
        u_star = u_star + MinvJT @ lam
        if out is not None:
            out["lam"] = lam
    if out is not None:
        out["rows"] = rows
        if not rows:
            out["lam"] = np.zeros(0)
    if getattr(prm, "max_velocity", 0.0) > 0.0:
        u_star = np.clip(u_star, -prm.max_velocity, prm.max_velocity)
    s.v, s.w, s.qd = u_star[0:3].copy(), u_star[3:6].copy(), u_star[6:].copy()
    integrate_positions(s, prm.dt)
    return touching


def energy(m, s):
    M, _, kin, _ = mass_matrix_and_bias(m, s)
    u = s.u()
    pot = -sum(m.body_mass[b] * GRAVITY @ kin["c"][b] for b in range(len(m.body_parent)))
    return 0.5 * u @ M @ u, pot


def momentum(m, s):
    """total linear momentum and angular momentum about the world origin."""
    kin = kinematics(m, s)
    u = s.u()
    P, Lw = np.zeros(3), np.zeros(3)
    for b in range(len(m.body_parent)):
        vc = point_jacobian(m, kin, b, kin["c"][b]) @ u
        wb = angular_jacobian(m, kin, b) @ u
        Iw = kin["R"][b] @ m.body_inertia[b] @ kin["R"][b].T
        P += m.body_mass[b] * vc
        Lw += np.cross(kin["c"][b], m.body_mass[b] * vc) + Iw @ wb
    return P, Lw


# ---- the pinned Python side: metalocomotion/envs/utils/walker_base.py, walker_base_env.py --------

HUMANOID_MOTOR_POWER = np.array([100, 100, 100, 100, 100, 300, 200, 100, 100, 300, 200, 75, 75, 75, 75, 75, 75], float)


def rot_to_rpy(R):
    """pybullet.getEulerFromQuaternion convention (XYZ fixed-axis roll, pitch, yaw)."""
    sy = -R[2, 0]
    pitch = np.arcsin(np.clip(sy, -1.0, 1.0))
    roll = np.arctan2(R[2, 1], R[2, 2])
    yaw = np.arctan2(R[1, 0], R[0, 0])
    return roll, pitch, yaw


def part_weights(m):
    """How many entries of `robot.parts` report body b's frame (walker_base.py:39-41 averages over ALL parts):
    the base once, every other body once per hinge joint it carries (a body with k joints is k links, the k-1
    massless intermediates reporting the same frame) and once if it has none (it hangs on a fixed joint) — the
    enumeration oracle/refstubs/pybullet attributes to Bullet's MJCF importer."""
    w = np.ones(len(m.body_parent))
    for b in range(1, len(m.body_parent)):
        w[b] = max(1, int(np.count_nonzero(np.asarray(m.joint_body) == b)))
    return w


class WalkerEnv(object):
    """calc_state / step bookkeeping of WalkerBase + WalkerBaseEnv around `substep`.

    torque_f32: Humanoid.apply_action (humanoids.py:50-54) multiplies python floats into np.clip(a[i], -1, +1),
    which for the float32 actions a gym Box hands over is a float32 scalar: under NumPy-2 promotion the whole
    product `1 * power * 0.41 * clip` is evaluated in float32. WalkerBase.apply_action (walker_base.py:26-29, the
    ant) converts with float() first and multiplies in float64.
    initial_z: 0.8 for the humanoid (a python float, humanoids.py:48: the alive test `state[0] + initial_z` is
    then a float32 sum); None for the ant — taken from the first calc_state (walker_base.py:44-45), a float64."""

    def __init__(self, model, prm=None, motor_power=HUMANOID_MOTOR_POWER, alive_z=0.50, alive_bonus=2.0,
                 max_steps=2000, initial_z=0.8, floor_in_parts=True, torque_f32=True):
        self.m, self.prm = model, prm or Params()
        self.motor_power, self.alive_z, self.alive_bonus = motor_power, alive_z, alive_bonus
        self.max_steps, self.initial_z_cfg = max_steps, initial_z
        # WalkerBaseEnv.reset re-runs addToScene on the ground bodies AFTER robot.reset(), so the floor link (at
        # the origin) joins robot.parts (walker_base_env.py:30-31) from the first step on — and stays there for
        # every later reset() of the same robot object, i.e. until the next set_task()
        self.floor_in_parts = floor_in_parts
        self.floor_known = False
        self.torque_f32 = torque_f32
        self.weights = part_weights(model)
        self.walk_target = np.array([1e3, 0.0])

    def reset(self, joint_noise):
        s = State(self.m)
        s.q = np.asarray(joint_noise, float).copy()       # walker_base.py:15
        self.s = s
        self.steps = 0
        self.feet_contact = np.zeros(len(self.m.foot_body))
        self.initial_z = self.initial_z_cfg               # humanoids.py:48 / walker_base.py:24
        obs = self.calc_state()
        self.potential = self.calc_potential()            # env_bases.py:80, before the floor joins the parts
        self.floor_known = self.floor_in_parts            # walker_base_env.py:30-31
        return obs

    def calc_state(self):
        m, s = self.m, self.s
        kin = kinematics(m, s)
        lo, hi = m.joint_lo, m.joint_hi
        pos = 2 * (s.q - 0.5 * (lo + hi)) / (hi - lo)                      # robot_bases.py:317-323
        vel = 0.1 * s.qd                                                    # :327-328 (revolute)
        j = np.stack([pos, vel], 1).astype(np.float32).flatten()
        self.joints_at_limit = int(np.count_nonzero(np.abs(j[0::2]) > 0.99))
        cnt = self.weights.sum() + (1.0 if self.floor_known else 0.0)      # the floor link sits at the origin
        self.body_xyz = (float(self.weights @ kin["o"][:, 0]) / cnt, float(self.weights @ kin["o"][:, 1]) / cnt,
                         kin["o"][0][2])
        roll, pitch, yaw = rot_to_rpy(kin["R"][0])
        self.body_rpy = (roll, pitch, yaw)
        z = self.body_xyz[2]
        if self.initial_z is None:
            self.initial_z = z                                              # walker_base.py:44-45
        theta = np.arctan2(self.walk_target[1] - self.body_xyz[1], self.walk_target[0] - self.body_xyz[0])
        self.walk_target_dist = np.linalg.norm([self.walk_target[1] - self.body_xyz[1],
                                                self.walk_target[0] - self.body_xyz[0]])
        ang = theta - yaw
        c, sn = np.cos(-yaw), np.sin(-yaw)
        rot_speed = np.array([[c, -sn, 0], [sn, c, 0], [0, 0, 1]])
        vx, vy, vz = rot_speed @ s.v
        more = np.array([z - self.initial_z, np.sin(ang), np.cos(ang), 0.3 * vx, 0.3 * vy, 0.3 * vz, roll, pitch],
                        dtype=np.float32)
        return np.clip(np.concatenate([more, j, self.feet_contact.astype(np.float32)]), -5, +5)

    def calc_potential(self):
        return -self.walk_target_dist / (self.prm.dt * self.prm.substeps)   # walker_base.py:66-82

    def torques(self, action):
        a32 = np.clip(np.asarray(action, np.float32), np.float32(-1), np.float32(1))
        if self.torque_f32:                                                  # humanoids.py:50-54, float32 product
            gain = (np.asarray(self.motor_power, float) * self.prm.power).astype(np.float32)
            return (gain * a32).astype(np.float64)
        return (self.prm.power * np.asarray(self.motor_power, float)) * a32.astype(np.float64)   # walker_base.py:26-29

    def step(self, action):
        m, prm = self.m, self.prm
        tau = self.torques(action)
        touching = set()
        for _ in range(prm.substeps):
            touching = substep(m, self.s, tau, prm)
        # the reference computes the state BEFORE refreshing feet_contact (walker_base_env.py:46 vs
        # :57-63), so the observation carries the previous step's contact flags
        state = self.calc_state()
        sph_foot = getattr(m, "sph_foot", None)                              # URDF robots: the proxy's own LINK decides
        for i, fb in enumerate(m.foot_body):                                 # walker_base_env.py:57-63
            if sph_foot is None:
                self.feet_contact[i] = 1.0 if any(m.sph_body[g] == fb for g in touching) else 0.0
            else:
                self.feet_contact[i] = 1.0 if any(sph_foot[g] == i for g in touching) else 0.0
        self.bad_contacts = 0 if sph_foot is None else sum(1 for g in touching if sph_foot[g] < 0)     # a1.py:314-323
        # walker_base_env.py:47: alive_bonus(state[0] + initial_z, ...) — float32 + python float stays float32
        # (humanoid), float32 + numpy float64 is float64 (ant: initial_z came out of calc_state)
        if self.initial_z_cfg is not None:
            height = float(np.float32(state[0]) + np.float32(self.initial_z))
        else:
            height = float(state[0]) + float(self.initial_z)
        alive = float(self.alive_bonus if height > self.alive_z else -1)
        done = alive < 0 or not np.isfinite(state).all()
        old = self.potential
        self.potential = self.calc_potential()
        progress = float(self.potential - old)
        rewards = [alive, progress, 0.0, float(-0.1 * self.joints_at_limit), 0.0]
        self.steps += 1
        done = bool(done) or self.steps >= self.max_steps
        return state, sum(rewards), done, {"rewards": rewards, "steps": self.steps}
