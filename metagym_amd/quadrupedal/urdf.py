"""URDF -> flat articulated-body model for the engine (host side, runs once per robot file).

The reference hands `a1/a1.urdf` to PyBullet (`loadURDF`, metagym/quadrupedal/robots/a1.py:266-277; the file ships with
pybullet_data, not with the reference). This module reads the URDF subset such robot files use and produces the same
`Model` the MJCF loader does (metagym_amd/metalocomotion/mjcf.py), so the articulated-body engine
(metagym_amd/csrc/walker.hip) runs it unchanged:

  * links with `<inertial>` (origin xyz / rpy, mass, full inertia tensor), any number of `<collision>` elements with
    `box` / `cylinder` / `sphere` / `capsule` (PyBullet's extension) geometry, optional `<contact><lateral_friction>`;
  * joints `revolute` / `continuous` (one hinge each: origin, axis, limits, `<dynamics damping>`), `fixed`;
  * links hanging on FIXED joints are merged into the body they are welded to (composite mass / centre of mass / inertia
    — dynamically identical), but every link keeps its name, its frame inside that body (`link_frame`) and the collision
    proxies it brought, so "which LINK touches the ground" (a1.py:299-323: toe links vs everything else) survives.

What PyBullet's importer is known to do, and the options that restate it (parity with PyBullet itself is UNPINNED —
nothing here can be run against it):
  * `inertia="bullet_aabb"` (default): `loadURDF` WITHOUT `URDF_USE_INERTIA_FROM_FILE` — the reference's call — ignores the
    file's tensor and recomputes a diagonal one from the collision shapes: btCompoundShape::calculateLocalInertia takes
    the axis-aligned bounding box of the link's shapes in its inertial frame and returns the solid-box formula
    m/12 (ly^2 + lz^2, lx^2 + lz^2, lx^2 + ly^2) about the inertial origin. `inertia="file"` uses the URDF tensor.
  * lateral friction of a link without `<contact>` is 0.5 (URDFLinkContactInfo's default); Bullet multiplies the two
    bodies' coefficients.
  * a convex link shape against the plane / the terrain boxes: the engine's contacts are point proxies — sphere centres with
    their radius, and for polytopes their extreme points with radius 0: the 8 corners of a box, points on both rims of a
    cylinder. Exact for vertex-face contacts against the plane; edge-edge contacts with terrain boxes are not generated.
"""
import xml.etree.ElementTree as ET

import numpy as np

from ..metalocomotion.mjcf import Model

DEFAULT_LATERAL_FRICTION = 0.5      # URDFLinkContactInfo::m_lateralFriction
MAX_PROXIES = 128                   # MG_WALKER_MAX_SPHERES


def _floats(s, n):
    v = [float(x) for x in s.split()]
    assert len(v) == n, (s, n)
    return np.array(v)


def _sincos(a):
    """sin / cos with the quarter turns exact: URDF files write pi/2 as 1.5708 or 1.57079632679, and a box rotated by
    "90 degrees" should stay axis-aligned to the last bit."""
    k = np.round(a / (np.pi / 2))
    if abs(a - k * (np.pi / 2)) < 1e-4:
        return [(0.0, 1.0), (1.0, 0.0), (0.0, -1.0), (-1.0, 0.0)][int(k) % 4]
    return float(np.sin(a)), float(np.cos(a))


def rpy_to_mat(rpy):
    """URDF fixed-axis roll, pitch, yaw: R = Rz(yaw) Ry(pitch) Rx(roll)."""
    (sr, cr), (sp, cp), (sy, cy) = _sincos(rpy[0]), _sincos(rpy[1]), _sincos(rpy[2])
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _origin(elem):
    o = None if elem is None else elem.find("origin")
    if o is None:
        return np.eye(3), np.zeros(3)
    return rpy_to_mat(_floats(o.get("rpy", "0 0 0"), 3)), _floats(o.get("xyz", "0 0 0"), 3)


class _Shape(object):
    """One collision element in its link's frame: kind, rotation R, position p, half extents h (local axes; cylinder and
    capsule along local z: (r, r, half length) — the capsule's excludes the caps)."""

    def __init__(self, kind, R, p, h):
        self.kind, self.R, self.p, self.h = kind, R, p, np.asarray(h, float)

    def aabb_half(self, Rq):
        """Half extents of the bounding box in a frame whose axes are the columns of Rq (expressed in the link frame),
        the way Bullet's getAabb does it: |basis| times the local half extents."""
        h = self.h.copy()
        if self.kind == "capsule":
            h = np.array([h[0], h[0], h[0] + h[2]])
        elif self.kind == "sphere":
            h = np.array([h[0]] * 3)
        return np.abs(Rq.T @ self.R) @ h

    def proxies(self, rim_points):
        """[(position in the link frame, radius)]"""
        R, p, h = self.R, self.p, self.h
        if self.kind == "sphere":
            return [(p.copy(), float(h[0]))]
        if self.kind == "capsule":
            return [(p + R @ np.array([0.0, 0.0, -h[2]]), float(h[0])), (p + R @ np.array([0.0, 0.0, h[2]]), float(h[0]))]
        if self.kind == "box":
            return [(p + R @ (np.array([sx, sy, sz]) * h), 0.0) for sx in (-1.0, 1.0) for sy in (-1.0, 1.0) for sz in (-1.0, 1.0)]
        out = []                                                    # cylinder: both rims
        for z in (-h[2], h[2]):
            for k in range(rim_points):
                s, c = _sincos(2.0 * np.pi * k / rim_points)
                out.append((p + R @ np.array([h[0] * c, h[0] * s, z]), 0.0))
        return out


def _parse_link(elem, mesh):
    name = elem.get("name")
    inertial = elem.find("inertial")
    mass, Ri, pi, tensor = 0.0, np.eye(3), np.zeros(3), np.zeros((3, 3))
    if inertial is not None:
        Ri, pi = _origin(inertial)
        mass = float(inertial.find("mass").get("value"))
        it = inertial.find("inertia")
        if it is not None:
            g = lambda k: float(it.get(k, 0.0))
            tensor = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
    shapes = []
    for col in elem.findall("collision"):
        R, p = _origin(col)
        geo = col.find("geometry")
        kid = list(geo)[0]
        if kid.tag == "box":
            shapes.append(_Shape("box", R, p, 0.5 * _floats(kid.get("size"), 3)))
        elif kid.tag == "sphere":
            shapes.append(_Shape("sphere", R, p, [float(kid.get("radius"))] * 3))
        elif kid.tag in ("cylinder", "capsule"):
            r, l = float(kid.get("radius")), float(kid.get("length"))
            shapes.append(_Shape(kid.tag, R, p, [r, r, 0.5 * l]))
        elif kid.tag == "mesh" and mesh == "skip":
            continue
        else:
            raise ValueError("link %r: unsupported collision geometry <%s> (box, cylinder, sphere, capsule are; pass "
                             "mesh='skip' to drop mesh collisions)" % (name, kid.tag))
    friction = DEFAULT_LATERAL_FRICTION
    contact = elem.find("contact")
    if contact is not None and contact.find("lateral_friction") is not None:
        friction = float(contact.find("lateral_friction").get("value"))
    return dict(name=name, mass=mass, Ri=Ri, pi=pi, tensor=tensor, shapes=shapes, friction=friction)


def _link_inertia(link, mode):
    """Inertia tensor about the link's centre of mass, in the LINK frame."""
    Ri = link["Ri"]
    if mode == "file":
        return Ri @ link["tensor"] @ Ri.T
    if mode != "bullet_aabb":
        raise ValueError("inertia must be 'bullet_aabb' or 'file', got %r" % (mode,))
    if not link["shapes"] or link["mass"] == 0.0:
        return np.zeros((3, 3))
    lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
    for sh in link["shapes"]:
        c = Ri.T @ (sh.p - link["pi"])                      # shape centre in the inertial frame
        h = sh.aabb_half(Ri)
        lo, hi = np.minimum(lo, c - h), np.maximum(hi, c + h)
    l = hi - lo
    diag = link["mass"] / 12.0 * np.array([l[1] ** 2 + l[2] ** 2, l[0] ** 2 + l[2] ** 2, l[0] ** 2 + l[1] ** 2])
    return Ri @ np.diag(diag) @ Ri.T


def _link_disc(link):
    """btCollisionShape::getAngularMotionDisc() of the link's compound collision shape (in the link's INERTIAL frame, where Bullet
    keeps it): half the diagonal of the shapes' axis-aligned bounding box plus the distance of the box's centre from the origin.
    Bullet's contact-breaking margin of the link is 0.02 x this (metalocomotion.mjcf.angular_motion_discs)."""
    if not link["shapes"]:
        return 0.0
    Ri = link["Ri"]
    lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
    for sh in link["shapes"]:
        c = Ri.T @ (sh.p - link["pi"])
        h = sh.aabb_half(Ri)
        lo, hi = np.minimum(lo, c - h), np.maximum(hi, c + h)
    return float(0.5 * np.linalg.norm(hi - lo) + np.linalg.norm(0.5 * (hi + lo)))


def _link_principal(link, mode):
    """(diag[3], Rp): the link's local inertia DIAGONAL — what pybullet.getDynamicsInfo(...)[2] reports and
    changeDynamics(localInertiaDiagonal=) sets — and the rotation of its axes in the LINK frame, so that the tensor of
    _link_inertia is Rp diag Rp^T. 'bullet_aabb': the bounding-box diagonal in the inertial frame; 'file': the URDF tensor's
    principal axes (Bullet diagonalises a full tensor on load)."""
    Ri = link["Ri"]
    if mode == "bullet_aabb":
        T = Ri.T @ _link_inertia(link, mode) @ Ri
        return np.diag(T).copy(), Ri
    w, V = np.linalg.eigh(link["tensor"])
    if np.allclose(link["tensor"], np.diag(np.diag(link["tensor"]))):
        w, V = np.diag(link["tensor"]).copy(), np.eye(3)
    return w, Ri @ V


def load_urdf(path_or_string, foot_links=(), inertia="bullet_aabb", armature=0.0, joint_order=None, rim_points="auto",
              mesh="error", root_pose=((0.0, 0.0, 0.0), None)):
    """-> `Model` (plus `link_names`, `link_body`, `link_frame`, `sph_link`, `sph_foot`, `sph_friction`, `joint_effort`,
    `joint_velocity`, `joint_friction`).

    foot_links   names of the links whose ground contact sets feet_contact (in this order); every other link's contact
                 points count as "bad" contacts
    inertia      'bullet_aabb' (what the reference's loadURDF call does) | 'file'
    armature     rotor inertia added to the diagonal of the joint-space inertia matrix (URDF has no such attribute; 0)
    joint_order  optional list of joint names: siblings are visited so that the model's hinge order follows it (the engine's
                 hinge order is depth-first document order otherwise) — e.g. a1.MOTOR_NAMES
    rim_points   contact proxies per cylinder rim; 'auto' = the most of (8, 6, 4, 3) that keeps the robot within 128 proxies
    root_pose    (xyz, R or None): where the root link sits at reset (Model.body_pos[0] / body_rot[0])
    """
    text = path_or_string
    if "<robot" not in text:
        with open(path_or_string, "r") as f:
            text = f.read()
    root = ET.fromstring(text)
    links = {l.get("name"): _parse_link(l, mesh) for l in root.findall("link")}
    link_order = [l.get("name") for l in root.findall("link")]
    joints = []
    for j in root.findall("joint"):
        t = j.get("type")
        if t not in ("revolute", "continuous", "fixed"):
            raise ValueError("joint %r: type %r is not supported (revolute, continuous, fixed)" % (j.get("name"), t))
        R, p = _origin(j)
        axis = _floats(j.find("axis").get("xyz"), 3) if j.find("axis") is not None else np.array([1.0, 0.0, 0.0])
        lim, dyn = j.find("limit"), j.find("dynamics")
        lo, hi = -1e30, 1e30
        if t == "revolute":
            lo, hi = float(lim.get("lower", 0.0)), float(lim.get("upper", 0.0))
        joints.append(dict(name=j.get("name"), type=t, parent=j.find("parent").get("link"), child=j.find("child").get("link"),
                           R=R, p=p, axis=axis / np.linalg.norm(axis), lo=lo, hi=hi,
                           effort=float(lim.get("effort", 0.0)) if lim is not None else 0.0,
                           velocity=float(lim.get("velocity", 0.0)) if lim is not None else 0.0,
                           damping=float(dyn.get("damping", 0.0)) if dyn is not None else 0.0,
                           friction=float(dyn.get("friction", 0.0)) if dyn is not None else 0.0))
    children = {}
    for j in joints:
        children.setdefault(j["parent"], []).append(j)
    is_child = {j["child"] for j in joints}
    roots = [n for n in link_order if n not in is_child]
    assert len(roots) == 1, "expected exactly one root link, got %r" % (roots,)

    if joint_order is not None:
        rank = {n: i for i, n in enumerate(joint_order)}

        def first_rank(j):
            best = rank.get(j["name"], len(rank)) if j["type"] != "fixed" else len(rank)
            for c in children.get(j["child"], []):
                best = min(best, first_rank(c))
            return best
        for k in children:
            children[k] = sorted(children[k], key=first_rank)       # stable: unranked siblings keep document order

    bodies, hinges, link_body, link_frame = [], [], {}, {}

    def add_link(name, b, R, p):
        """link `name` sits in body b with frame (R, p)"""
        link_body[name], link_frame[name] = b, (R, p)
        bodies[b]["links"].append(name)
        for j in children.get(name, []):
            Rc, pc = R @ j["R"], p + R @ j["p"]
            if j["type"] == "fixed":
                add_link(j["child"], b, Rc, pc)
            else:
                nb = len(bodies)
                bodies.append(dict(parent=b, pos=pc, rot=Rc, links=[], name=j["child"]))
                hinges.append(dict(j, body=nb))
                add_link(j["child"], nb, np.eye(3), np.zeros(3))

    xyz, R0 = root_pose
    bodies.append(dict(parent=-1, pos=np.asarray(xyz, float), rot=np.eye(3) if R0 is None else np.asarray(R0, float), links=[],
                       name=roots[0]))
    add_link(roots[0], 0, np.eye(3), np.zeros(3))

    n_cyl = sum(1 for l in links.values() for sh in l["shapes"] if sh.kind == "cylinder")
    n_other = sum(len(sh.proxies(1)) for l in links.values() for sh in l["shapes"] if sh.kind != "cylinder")
    if rim_points == "auto":
        fits = [k for k in (8, 6, 4, 3) if n_other + 2 * k * n_cyl <= MAX_PROXIES]
        if not fits:
            raise ValueError("the collision shapes need %d + %d x 2 x rim_points contact proxies; the engine holds %d"
                             % (n_other, n_cyl, MAX_PROXIES))
        rim_points = fits[0]

    m = Model()
    nb = len(bodies)
    mass, com, inert = np.zeros(nb), np.zeros((nb, 3)), np.zeros((nb, 3, 3))
    sph = []            # (body, pos, radius, link index, friction)
    for b, body in enumerate(bodies):
        parts = []
        for name in body["links"]:
            l = links[name]
            R, p = link_frame[name]
            c = p + R @ l["pi"]
            parts.append((l["mass"], c, R @ _link_inertia(l, inertia) @ R.T))
            for sh in l["shapes"]:
                for (x, r) in sh.proxies(rim_points):
                    sph.append((b, p + R @ x, r, link_order.index(name), l["friction"]))
        massive = [x for x in parts if x[0] > 0]
        if len(massive) == 1:                    # a single link: its own numbers, untouched by any arithmetic
            mass[b], com[b], inert[b] = massive[0][0], massive[0][1], massive[0][2]
            continue
        mass[b] = sum(x[0] for x in parts)
        if mass[b] > 0:
            com[b] = sum(x[0] * x[1] for x in parts) / mass[b]
        for mk, c, I in parts:
            d = c - com[b]
            inert[b] += I + mk * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
    if len(sph) > MAX_PROXIES:
        raise ValueError("%d contact proxies; the engine holds %d" % (len(sph), MAX_PROXIES))
    m.body_names = [b["name"] for b in bodies]
    m.joint_names = [h["name"] for h in hinges]
    m.body_parent = np.array([b["parent"] for b in bodies], np.int32)
    m.body_pos = np.array([b["pos"] for b in bodies])
    m.body_rot = np.array([b["rot"] for b in bodies])
    m.body_mass, m.body_com, m.body_inertia = mass, com, inert
    nj = len(hinges)
    m.joint_body = np.array([h["body"] for h in hinges], np.int32)
    m.joint_anchor = np.zeros((nj, 3))                       # a URDF child frame sits on its joint
    m.joint_axis = np.array([h["axis"] for h in hinges]).reshape(nj, 3)
    m.joint_lo = np.array([h["lo"] for h in hinges])
    m.joint_hi = np.array([h["hi"] for h in hinges])
    m.joint_armature = np.full(nj, float(armature))
    m.joint_damping = np.array([h["damping"] for h in hinges])
    m.joint_stiffness = np.zeros(nj)
    m.joint_effort = np.array([h["effort"] for h in hinges])
    m.joint_velocity = np.array([h["velocity"] for h in hinges])
    m.joint_friction = np.array([h["friction"] for h in hinges])       # not modelled (the reference disables the joint motors)
    m.sph_body = np.array([s[0] for s in sph], np.int32)
    m.sph_pos = np.array([s[1] for s in sph]).reshape(len(sph), 3)
    m.sph_radius = np.array([s[2] for s in sph])
    m.sph_link = np.array([s[3] for s in sph], np.int32)
    m.sph_friction = np.array([s[4] for s in sph])
    m.sph_disc = np.array([_link_disc(links[link_order[s[3]]]) for s in sph])      # per proxy: its LINK's angular motion disc
    m.link_names = list(link_order)
    m.link_body = dict(link_body)
    m.link_frame = dict(link_frame)
    # every LINK's own inertial record inside the body it was merged into — what per-robot dynamics (a1_dynamics.py: masses and
    # local inertia diagonals rescaled per link, like changeDynamics does) recompose the bodies from: body index, mass, centre of
    # mass in the body frame, local inertia diagonal and its axes in the body frame
    m.link_parts = {}
    for name in link_order:
        l, (R, p) = links[name], link_frame[name]
        diag, Rp = _link_principal(l, inertia)
        m.link_parts[name] = dict(body=link_body[name], mass=float(l["mass"]), com=p + R @ l["pi"], diag=diag, axes=R @ Rp)
    # the URDF's joints (name, child link) in the order PyBullet numbers them — joint i carries link i: a pre-order walk of the tree
    # from the root, a link's child joints in document order (URDF2Bullet's ComputeParentIndices)
    m.urdf_joints = []

    def number(parent):
        for j in joints:
            if j["parent"] == parent:
                m.urdf_joints.append((j["name"], j["child"]))
                number(j["child"])
    number(roots[0])
    m.root_link = roots[0]
    m.foot_names = list(foot_links)
    for f in foot_links:
        assert f in links, "foot link %r is not in the URDF" % (f,)
    foot_idx = {link_order.index(f): i for i, f in enumerate(foot_links)}
    m.sph_foot = np.array([foot_idx.get(int(k), -1) for k in m.sph_link], np.int8)
    m.foot_body = np.array([link_body[f] for f in foot_links], np.int32)
    # no capsule geoms / self-collision pairs: the quadrupedal reference loads the robot with self collision off
    # (minitaur.py:83, a1.py:266-277) and boxes / cylinders are not capsules
    m.geom_body = np.zeros(0, np.int32)
    m.geom_p0, m.geom_p1, m.geom_radius = np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(0)
    m.pair_a, m.pair_b = np.zeros(0, np.int32), np.zeros(0, np.int32)
    m.geom_friction = np.array(float(np.mean(m.sph_friction)) if len(sph) else DEFAULT_LATERAL_FRICTION)
    m.rim_points = int(rim_points) if n_cyl else 0
    # PyBullet reports / resets the BASE at the root link's inertial frame origin (getBasePositionAndOrientation)
    m.root_inertial_pos = links[roots[0]]["pi"].copy()
    m.root_link_mass = float(links[roots[0]]["mass"])           # getDynamicsInfo(robot, -1)[0]: the base LINK's own mass
    assert all(m.body_parent[i] < i for i in range(nb)) and np.all(np.diff(m.joint_body) >= 0)
    return m
