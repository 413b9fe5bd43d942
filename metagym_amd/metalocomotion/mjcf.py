"""MJCF -> flat articulated-body model (host side, runs once per task file).

The reference hands the MJCF file to PyBullet (`loadMJCF`, metalocomotion/envs/utils/robot_bases.py:119),
whose importer is not available here; this module reads the subset of MJCF the MetaLocomotion
assets use (metalocomotion/envs/assets/humanoids/*.xml, ants/*.xml).

What the numbers in the file MEAN is a choice of `preset` (PRESETS below; DESIGN.md §3.4 has the table with the
evidence for each row):

  * `preset="bullet"` (DEFAULT) — the closest known reading of what the reference's backend does with the file:
    link inertia recomputed from the collision shapes by the bounding-box rule (loadMJCF is called without
    URDF_USE_INERTIA_FROM_FILE), the inertial frame left at the body origin, and the joint attributes `armature`,
    `damping`, `stiffness` ignored: Bullet's MJCF importer reads a joint's type, axis, position, range and `limited`
    only, and its joint record has no rotor-inertia or spring field at all. (Physical evidence for dropping `damping`
    together with `armature`: ants/ant.xml:8 gives every ant hinge `armature="1" damping="1"`; without the armature an
    ant leg's joint-space inertia is 0.0019 kg m^2, on which an explicit damping of 1 N m s diverges at any step above
    3.8 ms — 5 ms here, 4.2 ms in pybullet_envs, whose ant runs on the same file.) The world adds btMultiBody's 0.04 / 0.04 linear / angular velocity damping of every link (PyBullet's
    default, which MetaLocomotion never changes) and its clamp of every generalized velocity to +-100
    (m_maxCoordinateVelocity; without it the ant — 0.07 kg legs, 250 N m motors, `armature="1"` gone — overflows in five
    steps) — `Model.body_damping`, `Model.max_velocity`, picked up by the envs.
  * `preset="mujoco"` — MuJoCo's documented semantics of the same attributes (below); what rounds 1-3 of this
    repository ran, and what tests/golden/walker_rules.npz was recorded on.

Neither can be compared with PyBullet here (parity of this path is unpinned). MuJoCo's documented semantics:

  * `<compiler angle="degree" inertiafromgeom="true">`: joint ranges are degrees; masses and inertia
    come from the geoms at the default density 1000 kg/m^3 (exact solid capsule / sphere formulas).
  * one `<default>` block without classes: attribute defaults for `joint` and `geom`.
  * bodies form a tree; the first worldbody child is the floating base (its `free` joint is
    commented out in the assets, PyBullet makes the root body floating anyway).
  * a body may carry several hinge joints: they act in document order, each later joint's anchor
    and axis being carried along by the earlier ones (MuJoCo kinematic-tree semantics).
  * geoms: `capsule` with `fromto` + radius `size[0]`, `sphere` with `pos` + radius.

Output: `Model` — plain numpy arrays, body/joint order = document (depth-first) order, which is
also the order PyBullet enumerates links and therefore the order of `ordered_joints`
(robot_bases.py:54-95) that fixes the observation layout (walker_base.py:32).
"""
import xml.etree.ElementTree as ET

import numpy as np

DENSITY = 1000.0     # MuJoCo default geom density


def _floats(s, n=None):
    v = [float(x) for x in s.split()]
    assert n is None or len(v) == n, (s, n)
    return v


def _quat_to_mat(q):
    w, x, y, z = np.asarray(q, float) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _sphere_inertia(r, density=DENSITY):
    m = density * 4.0 / 3.0 * np.pi * r ** 3
    return m, np.eye(3) * (0.4 * m * r * r)


def _capsule_inertia(p0, p1, r, density=DENSITY):
    """Solid capsule (cylinder + two hemispherical caps). Returns mass, com, inertia about com
    in the frame the endpoints are given in."""
    p0, p1 = np.asarray(p0, float), np.asarray(p1, float)
    h = np.linalg.norm(p1 - p0)
    mc = density * np.pi * r * r * h
    mh = density * 2.0 / 3.0 * np.pi * r ** 3          # one hemisphere
    m = mc + 2 * mh
    i_ax = 0.5 * mc * r * r + 2 * (0.4 * mh * r * r)
    d = h / 2 + 3.0 * r / 8.0                           # hemisphere com from the capsule centre
    i_tr = mc * (3 * r * r + h * h) / 12.0 + 2 * ((83.0 / 320.0) * mh * r * r + mh * d * d)
    z = (p1 - p0) / h if h > 0 else np.array([0.0, 0.0, 1.0])
    zz = np.outer(z, z)
    inertia = i_ax * zz + i_tr * (np.eye(3) - zz)
    return m, 0.5 * (p0 + p1), inertia


class Model(object):
    """Flat arrays describing one robot. nb bodies, nj hinge joints, ng collision spheres.

    body_parent[nb]   parent body index (-1 for the floating base)
    body_pos[nb,3], body_rot[nb,3,3]   body frame in the parent body frame at zero joint angles
    body_mass[nb], body_com[nb,3] (body frame), body_inertia[nb,3,3] (about the com, body frame)
    joint_body[nj]    body the joint belongs to (joints of a body are consecutive, in order)
    joint_anchor[nj,3], joint_axis[nj,3]   in the body frame; axis normalised
    joint_lo/hi[nj] (rad), joint_armature/damping/stiffness[nj]
    geom_friction     lateral friction coefficient of the robot's geoms (first entry of `friction`)
    sph_body[ng], sph_pos[ng,3] (body frame), sph_radius[ng]   collision proxies: every sphere geom
        and both end caps of every capsule geom (a capsule touches a plane at its caps)
    geom_body/p0/p1/radius[ngeo]   the geoms as capsules (sphere: p0 == p1), body frame
    pair_a/pair_b[npair]           geom pairs tested for self-collision (bodies distinct and not
                                   ancestor-related: PyBullet's EXCLUDE_ALL_PARENTS rule)
    """

    def __init__(self):
        self.body_names, self.joint_names, self.geom_names = [], [], []

    def dof(self):
        return 6 + len(self.joint_names)

    def to_dict(self):
        d = {k: np.asarray(v) for k, v in self.__dict__.items() if isinstance(v, np.ndarray)}
        d["body_names"] = np.asarray(self.body_names)
        d["joint_names"] = np.asarray(self.joint_names)
        d["foot_names"] = np.asarray(self.foot_names)
        return d

    @classmethod
    def from_dict(cls, d):
        m = cls()
        for k in d:
            v = d[k]
            if k in ("body_names", "joint_names", "foot_names"):
                setattr(m, k, [str(x) for x in v])
            else:
                setattr(m, k, np.asarray(v))
        return m


def rest_lowest_point(m):
    """z of the lowest point of any collision proxy (sphere surface) with the robot in its reset pose: base at body_pos[0] /
    body_rot[0], every joint at zero. For the reference's variant humanoids this is NEGATIVE: gen_variant_humanoids.py:44 lowers
    the torso by 0.20 while the legs reach further down, so every episode starts with the feet ~9 cm inside the floor."""
    nb = len(m.body_parent)
    o, R = np.zeros((nb, 3)), np.zeros((nb, 3, 3))
    for b in range(nb):                      # parents come first (check_walker requires body_parent[b] < b)
        pos, rot = np.asarray(m.body_pos[b], np.float64), np.asarray(m.body_rot[b], np.float64).reshape(3, 3)
        p = int(m.body_parent[b])
        if p < 0:
            o[b], R[b] = pos, rot
        else:
            o[b], R[b] = o[p] + R[p] @ pos, R[p] @ rot
    z = [(o[int(b)] + R[int(b)] @ np.asarray(c, np.float64))[2] - float(r) for b, c, r in zip(m.sph_body, m.sph_pos, m.sph_radius)]
    return float(min(z))


CONTACT_BREAKING_THRESHOLD = 0.02      # Bullet's gContactBreakingThreshold


def angular_motion_discs(m):
    """Per collision proxy: btCollisionShape::getAngularMotionDisc() of the LINK the proxy belongs to — the radius of the bounding
    sphere Bullet derives from the axis-aligned bounding box of the link's (compound) collision shape, plus the distance of that
    box's centre from the shape's origin. Needed because Bullet's contact margin is RELATIVE by default:
    btCollisionDispatcher is constructed with CD_USE_RELATIVE_CONTACT_BREAKING_THRESHOLD, so a manifold's breaking threshold is
    min over the two shapes of gContactBreakingThreshold (0.02) x getAngularMotionDisc() — 2 % of the link's size, not 2 cm (the
    ground plane's own disc is huge and never the minimum). A Model that carries `sph_disc` (the URDF loader computes it per
    original link, in the link's inertial frame) is taken at its word; for MJCF models the link is the body, its capsule / sphere
    geoms in the body frame (the `bullet` preset leaves the inertial frame at the body origin)."""
    if hasattr(m, "sph_disc"):
        return np.asarray(m.sph_disc, np.float64)
    nb = len(m.body_parent)
    lo, hi = np.full((nb, 3), np.inf), np.full((nb, 3), -np.inf)
    for b, p0, p1, r in zip(m.geom_body, m.geom_p0, m.geom_p1, m.geom_radius):
        lo[b] = np.minimum(lo[b], np.minimum(p0, p1) - r)
        hi[b] = np.maximum(hi[b], np.maximum(p0, p1) + r)
    disc = np.zeros(nb)
    for b in range(nb):
        if np.all(np.isfinite(lo[b])):
            disc[b] = 0.5 * np.linalg.norm(hi[b] - lo[b]) + np.linalg.norm(0.5 * (hi[b] + lo[b]))
    return disc[np.asarray(m.sph_body, int)]


def contact_margins(m, contact_margin):
    """Per-proxy contact margin in metres for `contact_margin` = a float (that margin for every proxy; 0 = penetration only) or
    "relative" (Bullet's default rule: CONTACT_BREAKING_THRESHOLD x the link's angular motion disc)."""
    ns = len(m.sph_body)
    if isinstance(contact_margin, str):
        if contact_margin != "relative":
            raise ValueError("contact_margin must be a length in metres or 'relative', got %r" % (contact_margin,))
        return CONTACT_BREAKING_THRESHOLD * angular_motion_discs(m)
    return np.full(ns, float(contact_margin))


def grounded(m, clearance=0.0):
    """A copy of `m` whose base starts `clearance` above the height at which its lowest collision proxy just touches z = 0 in the
    reset pose (negative: that far inside). The episode then begins STANDING on the floor instead of being ejected from it —
    the contact-rich variant of a MetaLocomotion workload (bench.py `C4_grounded_*`). Everything else of the model is shared."""
    import copy
    g = copy.copy(m)
    g.body_pos = np.array(m.body_pos, dtype=np.float64, copy=True)
    g.body_pos[0, 2] += float(clearance) - rest_lowest_point(m)
    return g


def _aabb_box_inertia(mass, geoms, com):
    """What btCompoundShape::calculateLocalInertia returns for a body's shapes: the solid-box formula on the extents of
    their axis-aligned bounding box (in the body's axes), m/12 (ly^2 + lz^2, lx^2 + lz^2, lx^2 + ly^2), diagonal."""
    lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
    for p0, p1, r in geoms:
        lo = np.minimum(lo, np.minimum(p0, p1) - r)
        hi = np.maximum(hi, np.maximum(p0, p1) + r)
    l = hi - lo
    return np.diag(mass / 12.0 * np.array([l[1] ** 2 + l[2] ** 2, l[0] ** 2 + l[2] ** 2, l[0] ** 2 + l[1] ** 2]))


# What each preset makes of the file; every entry can be overridden per call (load_mjcf keyword of the same name).
PRESETS = {
    "bullet": dict(inertia="bullet_box", com="body_origin", armature="ignore", damping="ignore", stiffness="ignore",
                   body_damping=(0.04, 0.04), max_velocity=100.0, contact_margin="relative"),
    "mujoco": dict(inertia="geom", com="geom", armature="diagonal", damping="explicit", stiffness="spring",
                   body_damping=(0.0, 0.0), max_velocity=0.0, contact_margin=0.0),
}
DEFAULT_PRESET = "bullet"


def preset_options(preset=None, **over):
    """The option dict of a preset with the non-None overrides applied."""
    preset = DEFAULT_PRESET if preset is None else preset
    if preset not in PRESETS:
        raise ValueError("unknown preset %r (%s)" % (preset, ", ".join(sorted(PRESETS))))
    opt = dict(PRESETS[preset])
    for k, v in over.items():
        if v is not None:
            opt[k] = v
    return opt


def load_mjcf(path_or_string, foot_names=("right_foot", "left_foot"), preset=None, inertia=None, com=None, armature=None,
              damping=None, stiffness=None):
    """preset: 'bullet' (default) | 'mujoco' — see the module docstring; the four keywords override single rows of it:
    inertia: 'geom' — exact solid capsule / sphere tensors at the geoms' density, MuJoCo's documented
    `inertiafromgeom`; 'bullet_box' — what PyBullet's loader does with a body's shapes when it is not told to trust the
    file (robot_bases.py:119 passes no URDF_USE_INERTIA_FROM_FILE): the solid-box formula on the bounding box of the
    body's geoms, diagonal in the body's axes (btCompoundShape::calculateLocalInertia); masses stay volume x density.
    com: 'geom' — mass-weighted geom centroid (MuJoCo); 'body_origin' — the body frame's origin, where Bullet's MJCF
    importer is believed to leave the inertial frame of a body without an <inertial> element (unverifiable here).
    armature: 'diagonal' — added to the diagonal of the joint-space inertia (MuJoCo); 'ignore' — dropped.
    damping: 'explicit' — torque -d qd evaluated at the start of the sub-step (MuJoCo's passive force, explicit here);
    'ignore' — dropped.
    stiffness: 'spring' — explicit torque -k q (MuJoCo, springref 0); 'ignore' — dropped."""
    opt = preset_options(preset, inertia=inertia, com=com, armature=armature, damping=damping, stiffness=stiffness)
    inertia, com, armature, damping, stiffness = opt["inertia"], opt["com"], opt["armature"], opt["damping"], opt["stiffness"]
    assert inertia in ("geom", "bullet_box") and com in ("geom", "body_origin")
    assert armature in ("diagonal", "ignore") and damping in ("explicit", "ignore") and stiffness in ("spring", "ignore")
    inertia_mode, com_mode = inertia, com
    text = path_or_string
    if "<mujoco" not in text:
        with open(path_or_string, "r") as f:
            text = f.read()
    root = ET.fromstring(text)
    comp = root.find("compiler")
    degrees = comp is None or comp.get("angle", "degree") == "degree"
    jdef, gdef = {}, {}
    dflt = root.find("default")
    if dflt is not None:
        if dflt.find("joint") is not None:
            jdef = dict(dflt.find("joint").attrib)
        if dflt.find("geom") is not None:
            gdef = dict(dflt.find("geom").attrib)

    bodies, joints, spheres, frictions, geoms = [], [], [], [], []

    def visit(elem, parent):
        b = len(bodies)
        pos = np.array(_floats(elem.get("pos", "0 0 0"), 3))
        rot = _quat_to_mat(_floats(elem.get("quat", "1 0 0 0"), 4))
        masses, coms, inertias, extents = [], [], [], []
        for g in elem.findall("geom"):
            a = dict(gdef)
            a.update(g.attrib)
            size = _floats(a["size"])
            density = float(a.get("density", DENSITY))
            frictions.append(_floats(a.get("friction", "1 0.005 0.0001"))[0])
            if a.get("type", "sphere") == "capsule":
                ft = _floats(a["fromto"], 6)
                m, c, inertia = _capsule_inertia(ft[:3], ft[3:], size[0], density)
                spheres.append((b, np.array(ft[:3]), size[0], a.get("name", "")))
                spheres.append((b, np.array(ft[3:]), size[0], a.get("name", "")))
                geoms.append((b, np.array(ft[:3]), np.array(ft[3:]), size[0]))
                extents.append((np.array(ft[:3]), np.array(ft[3:]), size[0]))
            elif a.get("type", "sphere") == "sphere":
                c = np.array(_floats(a.get("pos", "0 0 0"), 3))
                m, inertia = _sphere_inertia(size[0], density)
                spheres.append((b, c, size[0], a.get("name", "")))
                geoms.append((b, c, c, size[0]))
                extents.append((c, c, size[0]))
            else:
                raise ValueError("unsupported geom type %r" % a.get("type"))
            masses.append(m)
            coms.append(c)
            inertias.append(inertia)
        mass = float(sum(masses))
        com = sum(m * c for m, c in zip(masses, coms)) / mass if mass > 0 else np.zeros(3)
        inertia = np.zeros((3, 3))
        for m, c, i3 in zip(masses, coms, inertias):
            d = c - com
            inertia += i3 + m * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        if com_mode == "body_origin":
            com = np.zeros(3)
        if inertia_mode == "bullet_box" and mass > 0:
            inertia = _aabb_box_inertia(mass, extents, com)
        bodies.append(dict(name=elem.get("name", "body%d" % b), parent=parent, pos=pos, rot=rot, mass=mass, com=com,
                           inertia=inertia))
        for j in elem.findall("joint"):
            a = dict(jdef)
            a.update(j.attrib)
            if a.get("type", "hinge") != "hinge":
                raise ValueError("only hinge joints are supported (got %r)" % a.get("type"))
            axis = np.array(_floats(a.get("axis", "0 0 1"), 3))
            lo, hi = _floats(a.get("range", "0 0"), 2)
            if degrees:
                lo, hi = np.deg2rad(lo), np.deg2rad(hi)
            joints.append(dict(name=a.get("name", "joint%d" % len(joints)), body=b,
                               anchor=np.array(_floats(a.get("pos", "0 0 0"), 3)), axis=axis / np.linalg.norm(axis),
                               lo=lo, hi=hi, armature=float(a.get("armature", 0)), damping=float(a.get("damping", 0)),
                               stiffness=float(a.get("stiffness", 0))))
        for child in elem.findall("body"):
            visit(child, b)

    world = root.find("worldbody")
    top = world.findall("body")
    assert len(top) == 1, "expected exactly one root body"
    visit(top[0], -1)

    m = Model()
    m.body_names = [b["name"] for b in bodies]
    m.joint_names = [j["name"] for j in joints]
    m.body_parent = np.array([b["parent"] for b in bodies], np.int32)
    m.body_pos = np.array([b["pos"] for b in bodies])
    m.body_rot = np.array([b["rot"] for b in bodies])
    m.body_mass = np.array([b["mass"] for b in bodies])
    m.body_com = np.array([b["com"] for b in bodies])
    m.body_inertia = np.array([b["inertia"] for b in bodies])
    m.joint_body = np.array([j["body"] for j in joints], np.int32)
    m.joint_anchor = np.array([j["anchor"] for j in joints])
    m.joint_axis = np.array([j["axis"] for j in joints])
    m.joint_lo = np.array([j["lo"] for j in joints])
    m.joint_hi = np.array([j["hi"] for j in joints])
    m.joint_armature = np.array([j["armature"] if armature == "diagonal" else 0.0 for j in joints])
    m.joint_damping = np.array([j["damping"] if damping == "explicit" else 0.0 for j in joints])
    m.joint_stiffness = np.array([j["stiffness"] if stiffness == "spring" else 0.0 for j in joints])
    # the world-side half of the preset: btMultiBody's (linear, angular) velocity damping of every link; the envs take it
    # from their own `preset` / `body_damping` arguments, this copy records what the model was loaded for
    m.body_damping = np.array(opt["body_damping"], np.float64)
    m.max_velocity = np.array(float(opt["max_velocity"]))      # btMultiBody's m_maxCoordinateVelocity clamp (0 = off)
    m.contact_margin = np.array(opt["contact_margin"])         # the world's contact margin rule: metres, or "relative" (contact_margins)
    m.preset = np.array(DEFAULT_PRESET if preset is None else preset)
    m.sph_body = np.array([s[0] for s in spheres], np.int32)
    m.sph_pos = np.array([s[1] for s in spheres])
    m.sph_radius = np.array([s[2] for s in spheres])
    # capsule geoms (a sphere is a capsule with p0 == p1) for the self-collision narrow phase, and the
    # body pairs PyBullet's URDF_USE_SELF_COLLISION | URDF_USE_SELF_COLLISION_EXCLUDE_ALL_PARENTS
    # (robot_bases.py:119) leaves active: distinct bodies, neither an ancestor of the other
    m.geom_body = np.array([g[0] for g in geoms], np.int32)
    m.geom_p0 = np.array([g[1] for g in geoms])
    m.geom_p1 = np.array([g[2] for g in geoms])
    m.geom_radius = np.array([g[3] for g in geoms])

    def ancestors(b):
        out = set()
        while m.body_parent[b] >= 0:
            b = int(m.body_parent[b])
            out.add(b)
        return out
    anc = [ancestors(b) for b in range(len(bodies))]
    has_joint = [bool(np.any(m.joint_body == b)) for b in range(len(bodies))]

    def rigid_group(b):     # bodies welded together (no joint in between) can never move relative to each other
        while m.body_parent[b] >= 0 and not has_joint[b]:
            b = int(m.body_parent[b])
        return b
    grp = [rigid_group(b) for b in range(len(bodies))]
    pairs = [(ga, gb) for ga in range(len(geoms)) for gb in range(ga + 1, len(geoms))
             if grp[geoms[ga][0]] != grp[geoms[gb][0]] and geoms[ga][0] not in anc[geoms[gb][0]]
             and geoms[gb][0] not in anc[geoms[ga][0]]]
    m.pair_a = np.array([p[0] for p in pairs], np.int32)
    m.pair_b = np.array([p[1] for p in pairs], np.int32)
    m.geom_friction = np.array(float(np.mean(frictions)) if frictions else 1.0)   # lateral friction of the geoms
    m.foot_names = list(foot_names)
    m.foot_body = np.array([m.body_names.index(f) for f in foot_names], np.int32)
    # joints must be grouped by body in DFS order with parents before children
    assert all(m.body_parent[i] < i for i in range(len(bodies)))
    assert np.all(np.diff(m.joint_body) >= 0)
    return m
