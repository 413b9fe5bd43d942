"""Randomised dynamics of `quadrupedal-v0`, per robot and per reset, on the device.

`LocomotionGymEnv.reset` with `random_dynamic=True` (quadrupedal/envs/locomotion_gym_env.py:381-413) redraws, for its ONE robot,
from numpy's global stream and in this order

    control latency    U(0.035, 0.045) s                      SetControlLatency                  :382-383
    foot friction      U(1, 2)                                SetFootFriction (the toe links)    :384, :408
    base mass ratio    U(0.8, 1.2)                            SetBaseMasses([m_base * r])        :385-386, :409
    base inertia ratio U((0.3, 1.3, 0.5), (0.7, 1.7, 1.5))   SetBaseInertias: the three entries of the base link's local inertia
                                                              diagonal, one ratio each            :387-389, :410
    leg mass ratio     U((1, 1.1, 0.8), (1.4, 1.5, 1.6))      SetLegMasses(legmass * [r0, r1, r2] * 4)   :390-391, :411
    leg inertia ratio  U(lo[12], hi[12])                      SetLegInertias: link i's whole diagonal x ratio[i]   :392-400, :412
    motor kp           U(lo[12], hi[12]),  motor kd  N(mean[12], std[12])     SetMotorGains      :401-405
    gravity            U((-1, -1, 8), (1, 1, 12))             setGravity(g0, g1, g2)             :406-407

Two things in there are kept exactly as they are because they are what the reference does (replicate, do not
fix): (1) the twelve "leg" entries are `_leg_link_ids` (the lower and toe links, sorted by PyBullet link id: lower FR, toe FR,
lower FL, toe FL, ...) followed by `_motor_link_ids` (the four upper links) (minitaur.py:270-276, a1.py:396-430), so the pattern
[r0, r1, r2] x 4 does NOT line up with (hip, upper, lower): lower FR gets r0, toe FR r1, lower FL r2, toe FL r0, ...; (2) the z
component of the drawn gravity is POSITIVE and goes to setGravity unchanged: with random_dynamic the reference's gravity points up
(`gravity_sign=-1` below turns it down; not the default).

Here every robot of the batch draws its own set whenever IT is reset (full, masked or fused auto-reset) from a device generator
seeded by the env — same distributions, not numpy's numbers; `source` injects the draws (tests). Masses and inertias go where
Bullet's changeDynamics puts them: every LINK keeps its inertial frame, its mass and local inertia diagonal are rescaled, and the
engine's bodies (a URDF's fixed links are merged into the body they are welded to) are recomposed from their links, per robot, into
that robot's own row of the model table. `dynamic_param`'s baseinertia / legmass / leginertia keys (:360-373) take the same path with
given ratios."""
import re

import numpy as np
import torch

HIP, UPPER, LOWER, TOE, IMU = (re.compile(p) for p in (r"\w+_hip_\w+", r"\w+_upper_\w+", r"\w+_lower_\w+", r"\w+_toe\d*", r"imu\d*"))   # a1.py:75-79

# locomotion_gym_env.py:382-407, in draw order: name, kind, low / mean, high / std
RANGES = (("control_latency", "uniform", 0.035, 0.045),
          ("footfriction", "uniform", 1.0, 2.0),
          ("basemass_ratio", "uniform", 0.8, 1.2),
          ("baseinertia_ratio", "uniform", (0.3, 1.3, 0.5), (0.7, 1.7, 1.5)),
          ("legmass_ratio", "uniform", (1.0, 1.1, 0.8), (1.4, 1.5, 1.6)),
          ("leginertia_ratio", "uniform", (0.8, 0.55, 0.69, 1.4, 1.33, 0.5, 1.37, 0.48, 1.06, 1.39, 1.4, 1.4),
           (1.4, 0.75, 0.99, 1.6, 1.53, 0.8, 1.57, 0.68, 1.46, 1.59, 1.6, 1.6)),
          ("motor_kp", "uniform", (83, 89, 83, 65, 100, 73, 68, 78, 76, 67, 74, 66), (109, 109, 109, 109, 106, 97, 90, 80, 100, 99, 80, 109)),
          ("motor_kd", "normal", (1.1, 2.9, 2.15, 1.8, 3.19, 1.8, 1.1, 3.99, 1.7, 1.2, 3.99, 2.7),
           (1.9, 3.1, 2.85, 2.0, 3.21, 2.2, 1.9, 4.01, 2.1, 2.0, 4.01, 3.9)),
          ("gravity", "uniform", (-1.0, -1.0, 8.0), (1.0, 1.0, 12.0)))


def classify_links(urdf_joints, root_link):
    """a1.A1._BuildUrdfIds (a1.py:388-430) on the URDF's joints in document order (joint i carries link i, as PyBullet numbers
    them): -> (chassis link, the 12 "leg" links in the order of _leg_masses_urdf / _leg_inertia_urdf, the toe links)."""
    lower, toe, motor = [], [], []
    for i, (name, child) in enumerate(urdf_joints):
        if HIP.match(name):
            pass
        elif UPPER.match(name):
            motor.append((i, child))
        elif LOWER.match(name):
            lower.append((i, child))
        elif TOE.match(name):
            toe.append((i, child))
        elif IMU.match(name):
            pass
        else:
            raise ValueError("Unknown category of joint %s" % name)          # a1.py:420
    leg = sorted(lower + toe)                                                 # _leg_link_ids, sorted by link id
    return root_link, [c for _, c in leg] + [c for _, c in sorted(motor)], [c for _, c in sorted(toe)]


class A1Dynamics(object):
    """Per-robot dynamics on an `A1Physics` (which must have been built from a URDF: the links' inertial records are needed)."""

    def __init__(self, physics, seed=0, source=None, gravity_sign=1.0):
        m = physics.model
        if not hasattr(m, "link_parts"):
            raise ValueError("per-robot dynamics need a robot loaded from a URDF (Model.link_parts)")
        self.physics, self.n, self.device = physics, physics.n, physics.device
        self.source, self.gravity_sign = source, float(gravity_sign)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(seed) + 0xd1a)
        self.chassis, self.leg_links, self.toe_links = classify_links(m.urdf_joints, m.root_link)
        assert len(self.leg_links) == 12, "expected 4 x (lower, toe) + 4 upper links, got %r" % (self.leg_links,)
        f64 = dict(dtype=torch.float64, device=self.device)
        self._f64 = f64
        nb = len(m.body_parent)
        self.nb = nb
        # every massive link of the file: the body it was merged into, its nominal mass / local inertia diagonal, its inertial frame
        # in that body, and which ratio rescales it — column 0 of the ratio tables: 1, column 1: the base, 2 + i: leg entry i
        names = [n for n, p in m.link_parts.items() if p["mass"] > 0.0]
        parts = [m.link_parts[n] for n in names]
        self.link_names = names
        col = [1 if n == self.chassis else (2 + self.leg_links.index(n) if n in self.leg_links else 0) for n in names]
        self._col = torch.as_tensor(col, dtype=torch.long, device=self.device)
        self._body = torch.as_tensor([int(p["body"]) for p in parts], dtype=torch.long, device=self.device)
        self._lmass = torch.as_tensor(np.array([p["mass"] for p in parts], np.float64), **f64)
        self._lcom = torch.as_tensor(np.array([p["com"] for p in parts], np.float64), **f64)
        self._ldiag = torch.as_tensor(np.array([p["diag"] for p in parts], np.float64), **f64)
        self._laxes = torch.as_tensor(np.array([p["axes"] for p in parts], np.float64), **f64)
        onehot = np.zeros((len(names), nb))
        onehot[np.arange(len(names)), [int(p["body"]) for p in parts]] = 1.0
        self._onto = torch.as_tensor(onehot, **f64)                            # [L, nb]: sums links into bodies in a fixed order
        assert bool((self._onto.sum(0) > 0).all()), "a body without a massive link"
        self.base_mass_nominal = float(m.link_parts[self.chassis]["mass"])
        physics.enable_per_robot_dynamics()
        # what the env reads back: info["latency"], ["footfriction"], ["basemass"] (locomotion_gym_env.py:451-453)
        self.latency = torch.full((self.n,), 0.0, **f64)
        self.footfriction = torch.full((self.n,), 1.0, **f64)
        self.basemass = torch.full((self.n,), self.base_mass_nominal, **f64)
        self.motor_kp = torch.zeros(self.n, 12, **f64)
        self.motor_kd = torch.zeros(self.n, 12, **f64)
        self.last = None

    # ---- the draws ---------------------------------------------------------------------------------------------------
    def draw(self):
        """One set per robot: dict name -> [N] or [N, k] float64 tensors (every robot draws; the caller keeps the masked ones)."""
        if self.source is not None:       # injected draws: per name a scalar / [k] (the same for every robot) or [N] / [N, k]
            vals, out = self.source(), {}
            for name, kind, a, b in RANGES:
                k = len(a) if np.ndim(a) else 0
                v = torch.as_tensor(np.asarray(vals[name], np.float64), **self._f64)
                want = (self.n, k) if k else (self.n,)
                out[name] = v.expand(want).contiguous() if v.dim() <= (1 if k else 0) else v.reshape(want).contiguous()
            return self._signed(out)
        out = {}
        for name, kind, a, b in RANGES:
            k = len(a) if np.ndim(a) else 0
            shape = (self.n, k) if k else (self.n,)
            lo, hi = torch.as_tensor(a, **self._f64), torch.as_tensor(b, **self._f64)
            if kind == "uniform":
                out[name] = lo + (hi - lo) * torch.rand(shape, generator=self.gen, **self._f64)
            else:                                                           # np.random.normal(mean, std)
                out[name] = lo + hi * torch.randn(shape, generator=self.gen, **self._f64)
        return self._signed(out)

    def _signed(self, drawn):
        """`gravity_sign` acts on DRAWN sets only (the z in [8, 12] that locomotion_gym_env.py:406-407 hands to setGravity as it
        is): +1 keeps the reference's upward gravity, -1 turns it down. A set from fixed() / dynamic_param already says which way
        its gravity points (default (0, 0, -10)) and is installed unchanged — ADVICE r4: apply() used to flip those too, so
        gravity_sign=-1 with per-link dynamic_param made robots fall upward."""
        if self.gravity_sign != 1.0:
            g = drawn["gravity"].clone()
            g[:, 2] *= self.gravity_sign
            drawn["gravity"] = g
        return drawn

    def fixed(self, control_latency=None, footfriction=1.0, basemass=1.0, baseinertia=(1.0, 1.0, 1.0), legmass=(1.0, 1.0, 1.0),
              leginertia=(1.0,) * 12, motor_kp=None, motor_kd=None, gravity=(0.0, 0.0, -10.0)):
        """The `dynamic_param` route (locomotion_gym_env.py:349-380): given ratios instead of draws, the same for every robot."""
        t = lambda v, k=0: torch.as_tensor(np.asarray(v, np.float64), **self._f64).expand((self.n, k) if k else (self.n,)).contiguous()
        return dict(control_latency=None if control_latency is None else t(control_latency), footfriction=t(footfriction),
                    basemass_ratio=t(basemass), baseinertia_ratio=t(baseinertia, 3), legmass_ratio=t(legmass, 3),
                    leginertia_ratio=t(leginertia, 12), motor_kp=None if motor_kp is None else t(motor_kp, 12),
                    motor_kd=None if motor_kd is None else t(motor_kd, 12), gravity=t(gravity, 3))

    # ---- masses and inertias -> the robots' model rows ------------------------------------------------------------------
    def body_tables(self, v):
        """Per robot: body mass [N, nb], centre of mass [N, nb, 3], inertia about it [N, nb, 3, 3] recomposed from the links
        with the ratios of `v` applied the way SetBaseMasses / SetBaseInertias / SetLegMasses / SetLegInertias apply them."""
        N = self.n
        one = torch.ones(N, 1, **self._f64)
        leg_pattern = v["legmass_ratio"].repeat(1, 4)                        # [r0, r1, r2] * 4 against the 12 leg entries (:391)
        mass_ratio = torch.cat([one, v["basemass_ratio"].reshape(N, 1), leg_pattern], dim=1)[:, self._col]                    # [N, L]
        diag_ratio = torch.cat([one.reshape(N, 1, 1).expand(N, 1, 3), v["baseinertia_ratio"].reshape(N, 1, 3),
                                v["leginertia_ratio"].reshape(N, 12, 1).expand(N, 12, 3)], dim=1)[:, self._col]               # [N, L, 3]
        ml, dl = self._lmass * mass_ratio, self._ldiag * diag_ratio
        mass = ml @ self._onto                                               # [N, nb]
        com = torch.einsum("nlk,lb->nbk", ml.unsqueeze(-1) * self._lcom, self._onto) / mass.unsqueeze(-1)
        A = self._laxes
        I = torch.einsum("lij,nlj,lkj->nlik", A, dl, A)                      # A diag(d) A^T in the body frame, about the link's own centre
        d = self._lcom.unsqueeze(0) - com[:, self._body]                     # ... moved to the body's centre of mass
        eye = torch.eye(3, **self._f64)
        I = I + ml.reshape(N, -1, 1, 1) * ((d * d).sum(-1).reshape(N, -1, 1, 1) * eye - d.unsqueeze(-1) * d.unsqueeze(-2))
        inertia = torch.einsum("nlik,lb->nbik", I, self._onto)
        return mass, com, inertia

    def apply(self, v, mask=None):
        """Install the set `v` (draw() / fixed()) on the robots in `mask` (None: all): their model rows, foot friction and gravity in
        the engine; latency / gains / base mass are kept here for the env (`latency`, `motor_kp`, `motor_kd`, `basemass`)."""
        N = self.n
        m = torch.ones(N, dtype=torch.bool, device=self.device) if mask is None else torch.as_tensor(mask, device=self.device).bool()
        mass, com, inertia = self.body_tables(v)
        self.physics.write_body_tables(mass, com, inertia, m)
        self.physics.write_gravity(v["gravity"], m)       # as given: draw() has applied gravity_sign, fixed() sets never get it
        self.physics.write_foot_friction(v["footfriction"], m)
        sel = lambda new, old: torch.where(m.reshape([-1] + [1] * (old.dim() - 1)), new, old)
        if v.get("control_latency") is not None:
            self.latency.copy_(sel(v["control_latency"], self.latency))
        self.footfriction.copy_(sel(v["footfriction"], self.footfriction))
        self.basemass.copy_(sel(self.base_mass_nominal * v["basemass_ratio"], self.basemass))
        if v.get("motor_kp") is not None:
            self.motor_kp.copy_(sel(v["motor_kp"], self.motor_kp))
            self.motor_kd.copy_(sel(v["motor_kd"], self.motor_kd))
        self.last = v
