"""quadrupedal-v0's randomised dynamics, the host / torch side (metagym_amd/quadrupedal/a1_dynamics.py), on the CPU: which URDF
links are the reference's "base" and twelve "leg" entries, the draws' ranges, and the recomposition of the engine's bodies from
rescaled links. The engine side (per-robot model rows, gravity, foot friction) is tests/test_a1_dynamics_gpu.py."""
import numpy as np
import pytest
import torch
from scipy import stats

from metagym_amd.quadrupedal import load_urdf
from metagym_amd.quadrupedal import a1_dynamics as ad
from metagym_amd.quadrupedal.a1_actuators import MOTOR_NAMES
from urdf_fixture import A1_LIKE_TOES, a1_like_urdf


class _Physics(object):
    """What A1Dynamics needs from A1Physics, recording instead of writing to an engine."""

    def __init__(self, model, n):
        self.model, self.n, self.device = model, n, torch.device("cpu")
        self.foot_friction, self.calls = 1.0, []

    def enable_per_robot_dynamics(self):
        self.per_robot = True

    def write_body_tables(self, mass, com, inertia, mask):
        self.tables = (mass, com, inertia, mask.clone())

    def write_gravity(self, g, mask):
        self.gravity = g.clone()

    def write_foot_friction(self, mu, mask):
        self.mu = mu.clone()


def _model(shuffle=False, inertia="bullet_aabb"):
    return load_urdf(a1_like_urdf(shuffle_legs=shuffle), foot_links=A1_LIKE_TOES, joint_order=MOTOR_NAMES, inertia=inertia)


def test_leg_entries_follow_the_reference_id_order():
    """a1.py:388-430 + minitaur.py:264-276: leg entries = lower and toe links sorted by PyBullet link id, then the upper links;
    link ids = pre-order walk of the URDF tree. For the stock leg order FR, FL, RR, RL that is lower FR, toe FR, lower FL, ..."""
    m = _model()
    chassis, legs, toes = ad.classify_links(m.urdf_joints, m.root_link)
    assert chassis == "trunk"
    assert legs == ["FR_lower", "FR_toe", "FL_lower", "FL_toe", "RR_lower", "RR_toe", "RL_lower", "RL_toe",
                    "FR_upper", "FL_upper", "RR_upper", "RL_upper"]
    assert toes == ["FR_toe", "FL_toe", "RR_toe", "RL_toe"]
    # a file with its legs written in another order: PyBullet's ids — and with them which ratio a link gets — follow the file
    m2 = _model(shuffle=True)
    _, legs2, _ = ad.classify_links(m2.urdf_joints, m2.root_link)
    assert legs2[:2] == ["RL_lower", "RL_toe"] and legs2[8:] == ["RL_upper", "FR_upper", "RR_upper", "FL_upper"]
    with pytest.raises(ValueError, match="Unknown category"):
        ad.classify_links([("tail_joint", "tail")], "trunk")


@pytest.mark.parametrize("inertia", ["bullet_aabb", "file"])
def test_nominal_ratios_give_back_the_loaded_robot(inertia):
    m = _model(inertia=inertia)
    d = ad.A1Dynamics(_Physics(m, 3))
    mass, com, I = d.body_tables(d.fixed())
    assert np.allclose(mass[1].numpy(), m.body_mass, rtol=1e-14, atol=0)
    assert np.allclose(com[2].numpy(), m.body_com, rtol=0, atol=1e-16)
    assert np.allclose(I[0].numpy(), m.body_inertia, rtol=1e-12, atol=1e-15)


def test_ratios_land_on_the_links_the_reference_would_change():
    """One ratio at a time against a by-hand recomposition of the touched body from its links (numpy, straight from the URDF
    records): base mass and its three diagonal entries; leg mass pattern [r0, r1, r2] * 4 over the twelve leg ENTRIES (so lower FR
    gets r0, toe FR r1, lower FL r2, toe FL r0 ... and the uppers r2, r0, r1, r2); leg inertia entry i."""
    m = _model()
    n = 2
    d = ad.A1Dynamics(_Physics(m, n))
    parts = m.link_parts

    def body_by_hand(body, mass_of, diag_of):
        ls = [(k, p) for k, p in parts.items() if p["body"] == body and p["mass"] > 0]
        M = sum(mass_of(k, p) for k, p in ls)
        c = sum(mass_of(k, p) * p["com"] for k, p in ls) / M
        I = np.zeros((3, 3))
        for k, p in ls:
            dd = p["com"] - c
            I += p["axes"] @ np.diag(diag_of(k, p)) @ p["axes"].T + mass_of(k, p) * (dd @ dd * np.eye(3) - np.outer(dd, dd))
        return M, c, I

    legmass, leginertia = (1.3, 1.15, 0.9), tuple(1.0 + 0.05 * i for i in range(12))
    v = d.fixed(basemass=1.17, baseinertia=(0.4, 1.6, 0.9), legmass=legmass, leginertia=leginertia)
    mass, com, I = (t[1].numpy() for t in d.body_tables(v))
    _, legs, _ = ad.classify_links(m.urdf_joints, m.root_link)
    pattern = list(legmass) * 4

    def mass_of(k, p):
        return p["mass"] * (1.17 if k == "trunk" else pattern[legs.index(k)] if k in legs else 1.0)

    def diag_of(k, p):
        return p["diag"] * (np.array([0.4, 1.6, 0.9]) if k == "trunk" else leginertia[legs.index(k)] if k in legs else 1.0)
    for b in range(len(m.body_parent)):
        M, c, Ib = body_by_hand(b, mass_of, diag_of)
        assert abs(mass[b] - M) < 1e-15 and np.abs(com[b] - c).max() < 1e-16 and np.abs(I[b] - Ib).max() < 1e-17, b
    # the quirk, spelled out: the FL calf + toe body carries r2 (lower FL, entry 2) and r0 (toe FL, entry 3)
    b = parts["FL_lower"]["body"]
    assert parts["FL_toe"]["body"] == b
    assert abs(mass[b] - (parts["FL_lower"]["mass"] * legmass[2] + parts["FL_toe"]["mass"] * legmass[0])) < 1e-15
    # hips are nobody's entry: untouched
    hb = parts["FR_hip"]["body"]
    assert np.allclose(mass[hb], m.body_mass[hb], rtol=1e-15) and np.allclose(I[hb], m.body_inertia[hb], rtol=1e-12, atol=1e-18)


def test_draws_have_the_reference_distributions():
    """locomotion_gym_env.py:382-407: every component against its range with a Kolmogorov-Smirnov test at 8192 robots, kd against
    Normal(mean, std) (np.random.normal's second argument is the standard deviation — the reference passes what reads like an upper
    bound), gravity z POSITIVE in [8, 12]."""
    m = _model()
    n = 8192
    d = ad.A1Dynamics(_Physics(m, n), seed=3)
    v = d.draw()
    for name, kind, a, b in ad.RANGES:
        x = v[name].numpy().reshape(n, -1)
        a, b = np.atleast_1d(np.asarray(a, float)), np.atleast_1d(np.asarray(b, float))
        assert x.shape[1] == len(a), name
        for k in range(len(a)):
            if kind == "uniform":
                assert x[:, k].min() >= a[k] and x[:, k].max() <= b[k], (name, k)
                p = stats.kstest(x[:, k], "uniform", args=(a[k], b[k] - a[k])).pvalue
            else:
                p = stats.kstest(x[:, k], "norm", args=(a[k], b[k])).pvalue
            assert p > 1e-4, (name, k, p)
    assert float(v["gravity"][:, 2].min()) >= 8.0
    # different robots, different sets; a second draw differs from the first; the same seed repeats them
    assert float((v["motor_kp"][0] - v["motor_kp"][1]).abs().max()) > 0
    v2 = d.draw()
    assert float((v2["footfriction"] - v["footfriction"]).abs().max()) > 0
    again = ad.A1Dynamics(_Physics(m, n), seed=3).draw()
    assert all(torch.equal(again[k], v[k]) for k in v)


def test_apply_only_touches_the_masked_robots():
    m = _model()
    n = 6
    ph = _Physics(m, n)
    d = ad.A1Dynamics(ph, seed=0)
    v = d.draw()
    mask = torch.tensor([True, False, True, False, False, True])
    d.apply(v, mask)
    assert torch.equal(ph.tables[3], mask)
    assert torch.equal(d.latency[mask], v["control_latency"][mask]) and bool((d.latency[~mask] == 0).all())
    assert torch.equal(d.basemass[mask], (d.base_mass_nominal * v["basemass_ratio"])[mask])
    assert bool((d.basemass[~mask] == d.base_mass_nominal).all())
    assert torch.equal(d.motor_kd[mask], v["motor_kd"][mask]) and bool((d.motor_kd[~mask] == 0).all())
    assert torch.equal(ph.gravity, v["gravity"])                       # (the physics masks what it writes)
    # gravity_sign = -1: the DRAWN z turned downwards (not the reference's behaviour, offered for sane use) ...
    ph2 = _Physics(m, n)
    d2 = ad.A1Dynamics(ph2, seed=0, gravity_sign=-1.0)
    v2 = d2.draw()                                                     # same seed, same numbers
    assert torch.equal(v2["gravity"][:, 2], -v["gravity"][:, 2]) and torch.equal(v2["gravity"][:, :2], v["gravity"][:, :2])
    assert bool((v2["gravity"][:, 2] <= -8.0).all())
    d2.apply(v2, mask)
    assert torch.equal(ph2.gravity, v2["gravity"])
    # ... and ONLY the drawn one: a fixed() / dynamic_param set says which way its gravity points and is installed as given
    # (it used to be flipped too, so per-link dynamic_param + gravity_sign=-1 made robots fall upward)
    f = d2.fixed(legmass=(1.1, 1.0, 0.9))
    d2.apply(f)
    assert bool((ph2.gravity[:, 2] == -10.0).all()) and bool((ph2.gravity[:, :2] == 0.0).all())
    f = d2.fixed(gravity=(0.5, 0.0, -9.0))
    d2.apply(f)
    assert bool((ph2.gravity[:, 2] == -9.0).all()) and bool((ph2.gravity[:, 0] == 0.5).all())


def test_injected_draws_replace_the_generator():
    """`dynamics_source` (A1GymEnv) / `source` here: a callable returning the draws by name — scalars / [k] vectors for the whole batch
    or [N] / [N, k] per robot — in place of the device generator (the `force_source` pattern)."""
    m = _model()
    n = 5
    per_robot_kp = np.arange(n * 12, dtype=np.float64).reshape(n, 12) + 70.0
    vals = dict(control_latency=0.04, footfriction=np.linspace(1.0, 2.0, n), basemass_ratio=1.1, baseinertia_ratio=[0.5, 1.5, 1.0],
                legmass_ratio=[1.2, 1.3, 0.9], leginertia_ratio=np.linspace(0.9, 1.5, 12), motor_kp=per_robot_kp,
                motor_kd=np.full(12, 2.0), gravity=[0.3, -0.2, 9.0])
    ph = _Physics(m, n)
    d = ad.A1Dynamics(ph, source=lambda: vals)
    v = d.draw()
    assert v["control_latency"].shape == (n,) and float(v["control_latency"][3]) == 0.04
    assert torch.equal(v["footfriction"], torch.as_tensor(np.linspace(1.0, 2.0, n)))
    assert v["baseinertia_ratio"].shape == (n, 3) and v["leginertia_ratio"].shape == (n, 12) and v["gravity"].shape == (n, 3)
    assert torch.equal(v["motor_kp"], torch.as_tensor(per_robot_kp)) and float(v["motor_kd"][4, 7]) == 2.0
    d.apply(v)
    assert torch.equal(ph.mu, v["footfriction"]) and torch.equal(ph.gravity[2], torch.tensor([0.3, -0.2, 9.0], dtype=torch.float64))
    assert torch.allclose(d.basemass, torch.full((n,), 1.1 * d.base_mass_nominal, dtype=torch.float64), rtol=1e-15, atol=0)
