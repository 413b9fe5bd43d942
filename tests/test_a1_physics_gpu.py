"""GPU: SURVEY.md §8(f)-2's physics half — `quadrupedal-v0` from a URDF on the articulated-body engine (`A1Physics`),
without a caller-supplied simulator. What can be pinned is pinned: the URDF path against the MJCF path of the same body
(bit for bit), the engine against its numpy oracle on a terrain course (> 64 contact proxies, per-link friction, feet / bad
contact bookkeeping), the loader options against the oracle. Dynamics parity with PyBullet stays UNPINNED: the fixtures
are this repo's own (tests/urdf_fixture.py), the reference's a1.urdf ships with pybullet_data."""
import os

import numpy as np
import pytest
import torch

import metagym_amd
from metagym_amd.metalocomotion import MetaHumanoidEnv, variants
from metagym_amd.metalocomotion.mjcf import load_mjcf
from metagym_amd.quadrupedal import MOTOR_NAMES, A1Physics, load_urdf
from metagym_amd.quadrupedal.terrain import task_terrain
from oracle import abd
from urdf_fixture import A1_LIKE_TOES, a1_like_urdf, model_to_urdf

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def MARGINS(m):
    """A1Physics' default contact margins (CONTACT_MARGIN = "relative": Bullet's 0.02 x the link's angular motion disc) per proxy
    of the phys object's model `m`, for the oracle's Params."""
    from metagym_amd.metalocomotion.mjcf import contact_margins
    return contact_margins(m, "relative")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STANDIN_XML = os.path.join(ROOT, "examples", "a1_standin", "a1_standin.xml")
CALVES = ("FR_calf", "FL_calf", "RR_calf", "RL_calf")


@pytest.mark.parametrize("fused", [True, False])
def test_urdf_path_equals_mjcf_path_bit_for_bit(fused):
    """The stand-in body as MJCF (parsed by mjcf.py) and as the URDF tests/urdf_fixture.py writes from it (explicit inertias,
    capsule collisions, per-link friction; parsed by urdf.py): the two `A1Physics` must give the same closed-loop env —
    observations, rewards, torques and engine state bit for bit, on the flat ground and on a `task=` terrain."""
    n = 96
    w = np.tile([[0.02], [0.0], [0.015]], (1, 20)) * np.sin(np.linspace(0, 2 * np.pi, 20))
    for task in ("plane", "slopestair"):
        m = load_mjcf(STANDIN_XML, foot_names=CALVES, preset="mujoco")        # (its URDF twin carries the same armature 0.01)
        envs = []
        # (one absolute contact margin for both: the default "relative" rule measures each LINK's size — the MJCF body with its
        #  capsules about the body origin, the URDF link about its inertial frame — and the two loaders' numbers differ in the last
        #  digits, which is not what this test is about)
        for phys in (A1Physics(n, model=m, device=DEV, fused=fused, contact_margin=0.004),
                     A1Physics(n, urdf=model_to_urdf(m), device=DEV, fused=fused, foot_links=CALVES, inertia="file", armature=0.01,
                               contact_margin=0.004)):
            envs.append(metagym_amd.make("quadrupedal-v0", num_envs=n, physics=phys, device=DEV, ETG=1, ETG_w=w, ETG_b=np.zeros(3),
                                         task=task, control_latency=0.0057))
        o0, _ = envs[0].reset()
        o1, _ = envs[1].reset()
        assert torch.equal(o0, o1)
        rs = np.random.RandomState(3)
        for k in range(6):
            a = torch.as_tensor(rs.uniform(-0.2, 0.2, (n, 12)), device=DEV)
            r0, r1 = envs[0].step(a), envs[1].step(a)
            assert torch.equal(r0[0], r1[0]), "observation, step %d" % k
            assert torch.equal(r0[1], r1[1]) and torch.equal(r0[2], r1[2])
            assert torch.equal(envs[0].last_torques, envs[1].last_torques), "torques, step %d" % k
            for key in ("pos", "rot", "vel", "omega", "q", "qd", "feet_contact", "bad_contacts"):
                assert torch.equal(getattr(envs[0].physics.env, key), getattr(envs[1].physics.env, key)), key
        assert torch.isfinite(r0[0]).all() and float(r0[3]["real_contact"].sum()) > 0


def _oracle_boxes(spec):
    out = []
    for half, pos, (x, y, z, w), mu in spec:
        nq = np.sqrt(x * x + y * y + z * z + w * w)
        x, y, z, w = x / nq, y / nq, z / nq, w / nq
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        out.append((np.array(pos, float), R, np.array(half, float), float(mu)))
    return out


def test_urdf_robot_on_a_terrain_course_matches_the_oracle():
    """The A1-shaped URDF fixture (box / cylinder / sphere links, merged fixed links, 124 contact proxies = two lane chunks,
    per-link friction, bounding-box inertias) dropped along the reference's `slopestair` course and driven by torques, two
    robots rolled onto their sides so trunk / hip / thigh proxies touch too. Every one of 150 sub-steps is checked on its
    own: the oracle (oracle/abd.py) steps from the state the GPU had before the sub-step and must land on the GPU's state
    (1e-9: positions, velocities, joint angles and rates), with identical toe flags and bad-contact counts. (A 150-sub-step
    free-running comparison is not a test of the kernel: a contact that closes one sub-step earlier on one side — a 1e-12
    difference in a depth — moves the trajectories 1e-5 apart.)"""
    n = 8
    phys = A1Physics(n, urdf=a1_like_urdf(), device=DEV, foot_links=A1_LIKE_TOES)
    m = phys.model
    assert len(m.sph_body) == 124 and list(m.joint_names) == MOTOR_NAMES
    add_h, env_info, boxes = task_terrain("slopestair")
    phys.set_terrain(boxes, [0.0, 0.0, 0.28 + add_h])
    phys.reset(None)
    e = phys.env
    xs = np.array([0.0, 0.6, 1.2, 1.9, 2.6, 3.3, 0.3, 1.5])      # start platform, up-slope, top, down-stair
    rs = np.random.RandomState(0)
    pos = e.pos.cpu().numpy()
    pos[0] += xs
    pos[2] += 0.25
    rot = e.rot.cpu().numpy()
    for k in (6, 7):
        c, s_ = np.cos(1.3), np.sin(1.3)
        rot[:, k] = np.array([[1, 0, 0], [0, c, -s_], [0, s_, c]]).reshape(9)       # rolled onto its side
    e.pos.copy_(torch.as_tensor(pos))
    e.rot.copy_(torch.as_tensor(rot))
    prm = abd.Params(contact_margin=MARGINS(m), dt=0.002, substeps=1, iterations=23, erp=0.2, friction=5.0, sphere_friction=m.sph_friction,
                     self_collision=False, gravity=10.0, terrain=_oracle_boxes(boxes))
    target = np.array([0, 0.9, -1.8] * 4, float)
    log = torch.empty(1, 43, n, dtype=torch.float64, device=DEV)
    keys = ("pos", "rot", "vel", "omega", "q", "qd")
    worst, bad_seen, feet_seen, terrain_seen, force_seen = 0.0, 0, 0, 0, 0.0
    for t in range(150):
        st = {k: getattr(e, k).cpu().numpy() for k in keys}
        tau = np.clip(80.0 * (target[:, None] - st["q"]) - 1.5 * st["qd"] + rs.uniform(-2, 2, (12, n)), -33.5, 33.5)
        e.step_actuated(torch.as_tensor(tau, device=DEV), raw_torque=True, n_substeps=1, log=log)
        g = {k: getattr(e, k).cpu().numpy() for k in keys}
        feet, bad, force = e.feet_contact.cpu().numpy(), e.bad_contacts.cpu().numpy(), e.foot_force.cpu().numpy()
        for k in range(n):
            s = abd.State(m)
            s.pos, s.rot, s.v, s.w = st["pos"][:, k].copy(), st["rot"][:, k].reshape(3, 3).copy(), st["vel"][:, k].copy(), st["omega"][:, k].copy()
            s.q, s.qd = st["q"][:, k].copy(), st["qd"][:, k].copy()
            out = {}
            touching = abd.substep(m, s, tau[:, k], prm, out=out)
            terrain_seen += sum(1 for r in out["rows"] if r[2] == 0 and r[4] >= 0 and r[5][2] != 1.0)     # a tilted contact normal: a box
            d = max(np.abs(g["q"][:, k] - s.q).max(), np.abs(g["qd"][:, k] - s.qd).max(), np.abs(g["pos"][:, k] - s.pos).max(),
                    np.abs(g["vel"][:, k] - s.v).max(), np.abs(g["omega"][:, k] - s.w).max(), np.abs(g["rot"][:, k] - s.rot.reshape(9)).max())
            worst = max(worst, d)
            # (round-off of the two kinematics orderings — the kernel composes the chain's transforms as a scan, the oracle
            # body by body — amplified by contacts at box edges: 3e-9 seen, 3e-10 with the kernel's former body-by-body pass;
            # round 6: with the contact margin the step has more discrete decisions (a proxy entering the margin, the deepest of two
            # nearby boxes, clamps of speculative rows): next to one of them the ORACLE ITSELF turns a 1-ulp change of its input into
            # 1.4e-7 within one sub-step (scripts/probe_margin_sensitivity.py, profiles/r06/walker_margin_sensitivity.txt; 5e-10
            # without margin). 3.6e-8 / 2.6e-7 seen GPU vs oracle on the two terrain tests)
            assert d < 1e-6, (t, k, d)
            o_feet = [float(any(m.sph_foot[g_] == f for g_ in touching)) for f in range(4)]
            o_bad = sum(1 for g_ in touching if m.sph_foot[g_] < 0)
            assert list(feet[:, k]) == o_feet and int(bad[k]) == o_bad, (t, k)
            o_force = abd.foot_forces(m, out["rows"], out["lam"], prm.dt)
            assert np.allclose(force[:, k], o_force, rtol=1e-7, atol=1e-7), (t, k, force[:, k], o_force)      # newtons
            force_seen = max(force_seen, o_force.max())
            bad_seen += o_bad
            feet_seen += int(sum(o_feet))
    assert bad_seen > 50 and feet_seen > 500 and terrain_seen > 20            # toe, non-toe and tilted terrain-box contacts all happened
    assert force_seen > 20.0                                                  # (a 12.6 kg robot landing on its toes)
    print("a1-like URDF on slopestair: max one-sub-step |state diff| GPU vs oracle %.2e over 150 sub-steps x 8 robots; %d toe / %d bad "
          "contact points" % (worst, feet_seen, bad_seen))


def test_candidate_buffer_overflow_in_the_two_chunk_kernel_matches_the_oracle():
    """Round 6's candidate record / deepest-first selection in the SHAPE-GENERIC kernel with two 64-proxy chunks: A1-like robots
    laid on their sides / backs on a low platform next to the plain ground, with a flat 2 cm margin so that most of the 124
    proxies are candidates — more than the 48 the record holds (later ones are dropped, in the oracle alike), ground and terrain
    candidates of the same proxy, far more than the 12 the solver keeps. One sub-step at a time against oracle/abd.py from the
    GPU's own previous state: state 1e-6 (see the tolerance note in the single-course test), toe flags and bad-contact counts
    (which count EVERY proxy inside the margin, recorded or not) identical; the oracle's bookkeeping proves the overflow happened."""
    n = 8
    phys = A1Physics(n, urdf=a1_like_urdf(), device=DEV, foot_links=A1_LIKE_TOES, contact_margin=0.02)
    m = phys.model
    boxes = [([0.6, 0.6, 0.02], [0.0, 0.0, 0.02], [0.0, 0.0, 0.0, 1.0], 5.0)]          # a 4 cm platform under half of each robot
    phys.set_terrain(boxes, [0.0, 0.0, 0.3])
    phys.reset(None)
    e = phys.env
    pos, rot = e.pos.cpu().numpy(), e.rot.cpu().numpy()
    rs = np.random.RandomState(5)
    for k in range(n):
        ang = (np.pi / 2 if k % 2 == 0 else np.pi) + rs.uniform(-0.1, 0.1)           # on its side / on its back
        c, s_ = np.cos(ang), np.sin(ang)
        rot[:, k] = np.array([[1, 0, 0], [0, c, -s_], [0, s_, c]]).reshape(9)
        pos[:, k] = [0.45 + 0.05 * k, 0.0, 0.13 + 0.01 * (k % 3)]                     # straddling the platform's +x edge
    e.pos.copy_(torch.as_tensor(pos))
    e.rot.copy_(torch.as_tensor(rot))
    from metagym_amd.metalocomotion.mjcf import contact_margins
    prm = abd.Params(contact_margin=contact_margins(m, 0.02), dt=0.002, substeps=1, iterations=23, erp=0.2, friction=5.0,
                     sphere_friction=m.sph_friction, self_collision=False, gravity=10.0, terrain=_oracle_boxes(boxes), max_velocity=100.0)
    keys = ("pos", "rot", "vel", "omega", "q", "qd")
    log = torch.empty(1, 43, n, dtype=torch.float64, device=DEV)
    target = np.array([0, 0.9, -1.8] * 4, float)
    overflow, over_cap, both_kinds, worst = 0, 0, 0, 0.0
    for t in range(60):
        st = {k: getattr(e, k).cpu().numpy() for k in keys}
        tau = np.clip(40.0 * (target[:, None] - st["q"]) - 1.0 * st["qd"], -33.5, 33.5)
        e.step_actuated(torch.as_tensor(tau, device=DEV), raw_torque=True, n_substeps=1, log=log)
        g = {k: getattr(e, k).cpu().numpy() for k in keys}
        feet, bad = e.feet_contact.cpu().numpy(), e.bad_contacts.cpu().numpy()
        for k in range(n):
            s = abd.State(m)
            s.pos, s.rot, s.v, s.w = st["pos"][:, k].copy(), st["rot"][:, k].reshape(3, 3).copy(), st["vel"][:, k].copy(), st["omega"][:, k].copy()
            s.q, s.qd = st["q"][:, k].copy(), st["qd"][:, k].copy()
            kin = abd.kinematics(m, s)
            mg = prm.contact_margin
            n_ground = sum(1 for g_, b in enumerate(m.sph_body) if m.sph_radius[g_] - (kin["o"][b] + kin["R"][b] @ m.sph_pos[g_])[2] > -mg[g_])
            cands, touching0 = abd.contact_candidates(m, s, kin, prm)
            overflow += int(len(touching0) > len(cands) or len(cands) == abd.MAX_CANDIDATES)
            over_cap += int(len(cands) > 12)
            both_kinds += int(any(c["cat"] == 0 for c in cands) and any(c["cat"] == 1 for c in cands))
            touching = abd.substep(m, s, tau[:, k], prm)
            d = max(np.abs(g["q"][:, k] - s.q).max(), np.abs(g["qd"][:, k] - s.qd).max(), np.abs(g["pos"][:, k] - s.pos).max(),
                    np.abs(g["vel"][:, k] - s.v).max(), np.abs(g["omega"][:, k] - s.w).max())
            worst = max(worst, d)
            assert d < 1e-6, (t, k, d, len(cands), n_ground)
            assert list(feet[:, k]) == [float(any(m.sph_foot[g_] == f for g_ in touching)) for f in range(4)], (t, k)
            assert int(bad[k]) == sum(1 for g_ in touching if m.sph_foot[g_] < 0), (t, k)
    assert over_cap > 100 and both_kinds > 50, (over_cap, both_kinds)
    assert overflow > 20, overflow                                   # the 48-entry record was full in these sub-steps
    print("a1-like lying across a platform edge, flat 2 cm margin: worst one-sub-step |state diff| %.2e; sub-steps with > 12 candidates %d, "
          "with ground + terrain candidates %d, with the 48-entry record full %d" % (worst, over_cap, both_kinds, overflow))


def test_two_courses_in_one_batch_match_the_oracle():
    """Per-robot terrains (mg_walker_params.terrain_id): a terrain TABLE of three courses — the `slopestair` task course, a
    `special` stair / slope course built with reset(hardset=True)'s arguments, an empty one — and eight robots spread over them.
    Every sub-step is checked on its own against oracle/abd.py stepping the robot on ITS course (1e-8), with identical toe flags and
    bad-contact counts; robots 0-3 and 4-7 start at the same x, so the two halves differ only by what they stand on."""
    from metagym_amd.quadrupedal.terrain import upstair_terrain
    n = 8
    phys = A1Physics(n, urdf=a1_like_urdf(), device=DEV, foot_links=A1_LIKE_TOES)
    m = phys.model
    h_a, _, boxes_a = task_terrain("slopestair")
    h_b, _, boxes_b = upstair_terrain(stepwidth=0.3, slope=0.34, stepheight=0.07, mode="special",
                                      env_vecs=[[0, 0, 1, 0, 0, 0.08, 0.25], [0, 1, 0, 0, 0.34, 0, 0], [0, 0, 0, 0, 0, 0, 0]] * 3)
    phys.set_terrain_table(3, 96)
    phys.write_course(0, boxes_a)
    phys.write_course(1, boxes_b)                       # (course 2 stays empty: the plain ground)
    tid = np.array([0, 0, 0, 0, 1, 1, 1, 2], np.int32)
    phys.terrain_id.copy_(torch.as_tensor(tid))
    heights = np.array([h_a, h_b, 0.0])[tid]
    phys.set_reset_pose(np.stack([np.zeros(n), np.zeros(n), 0.28 + heights]))
    phys.reset(None)
    e = phys.env
    xs = np.array([0.0, 1.1, 2.2, 3.1, 0.0, 1.1, 2.2, 1.1])
    pos = e.pos.cpu().numpy()
    pos[0] += xs
    pos[2] += 0.3
    e.pos.copy_(torch.as_tensor(pos))
    courses = [_oracle_boxes(boxes_a), _oracle_boxes(boxes_b), []]
    prms = [abd.Params(contact_margin=MARGINS(m), dt=0.002, substeps=1, iterations=23, erp=0.2, friction=5.0, sphere_friction=m.sph_friction, self_collision=False,
                       gravity=10.0, terrain=c, max_velocity=100.0) for c in courses]
    target = np.array([0, 0.9, -1.8] * 4, float)
    rs = np.random.RandomState(1)
    keys = ("pos", "rot", "vel", "omega", "q", "qd")
    log = torch.empty(1, 43, n, dtype=torch.float64, device=DEV)
    worst, box_contacts = 0.0, np.zeros(3, int)
    for t in range(120):
        st = {k: getattr(e, k).cpu().numpy() for k in keys}
        tau = np.clip(80.0 * (target[:, None] - st["q"]) - 1.5 * st["qd"] + rs.uniform(-2, 2, (12, n)), -33.5, 33.5)
        e.step_actuated(torch.as_tensor(tau, device=DEV), raw_torque=True, n_substeps=1, log=log)
        g = {k: getattr(e, k).cpu().numpy() for k in keys}
        feet, bad = e.feet_contact.cpu().numpy(), e.bad_contacts.cpu().numpy()
        for k in range(n):
            s = abd.State(m)
            s.pos, s.rot, s.v, s.w = st["pos"][:, k].copy(), st["rot"][:, k].reshape(3, 3).copy(), st["vel"][:, k].copy(), st["omega"][:, k].copy()
            s.q, s.qd = st["q"][:, k].copy(), st["qd"][:, k].copy()
            out = {}
            touching = abd.substep(m, s, tau[:, k], prms[tid[k]], out=out)
            box_contacts[tid[k]] += sum(1 for r in out["rows"] if r[2] == 0 and r[4] >= 0 and (s.pos[2] > 0.4 or r[5][2] != 1.0))
            d = max(np.abs(g["q"][:, k] - s.q).max(), np.abs(g["qd"][:, k] - s.qd).max(), np.abs(g["pos"][:, k] - s.pos).max(),
                    np.abs(g["vel"][:, k] - s.v).max(), np.abs(g["omega"][:, k] - s.w).max())
            worst = max(worst, d)
            assert d < 1e-6, (t, k, d)         # (see the tolerance note in the single-course test)
            assert list(feet[:, k]) == [float(any(m.sph_foot[g_] == f for g_ in touching)) for f in range(4)], (t, k)
            assert int(bad[k]) == sum(1 for g_ in touching if m.sph_foot[g_] < 0), (t, k)
    zs = e.pos[2].cpu().numpy()
    assert abs(zs[1] - zs[5]) > 0.02 or abs(zs[2] - zs[6]) > 0.02      # same x, different course: different height under the feet
    print("two courses in one batch: max one-sub-step |state diff| GPU vs oracle %.2e; z of robots 1 / 5 (x = 1.1): %.3f / %.3f; "
          "2 / 6 (x = 2.2): %.3f / %.3f" % (worst, zs[1], zs[5], zs[2], zs[6]))


def test_quadrupedal_v0_new_terrain_and_heading_for_part_of_the_batch():
    """The batched form of the reference's "a new terrain task per episode" (reset(hardset=True, ...), locomotion_gym_env.py:297-301)
    and reset(yaw=, x_noise=) (:327-338) on the engine: an env with a terrain table; half the robots get a new course, a start
    heading and position noise through configure_reset(mask, ...) and restart INSIDE the next step (reset_mask); they then stand
    at their course's height, turned by their heading, spread over the noise range — the others are untouched."""
    n = 64
    env = metagym_amd.make("quadrupedal-v0", num_envs=n, urdf=a1_like_urdf(), device=DEV, task="slopestair", terrain_slots=3, seed=5)
    obs, info = env.reset()
    a = torch.zeros(n, 12, dtype=torch.float64, device=DEV)
    for _ in range(6):
        obs, reward, done, info = env.step(a)
    z_a = info["base"][:, 2].clone()
    assert float((z_a - (0.28 + env.add_height)).abs().max()) < 0.06 and int(env.terrain_id.max()) == 0
    m = torch.arange(n, device=DEV) % 2 == 1
    env.configure_reset(m, hardset=True, mode="downstair", stepwidth=0.28, slope=0.3, stepheight=0.06, env_vec=[], yaw=0.5, x_noise=True)
    assert sorted(set(env.terrain_id.cpu().tolist())) == [0, 1]
    untouched = {k: getattr(env.physics.env, k)[:, ~m].clone() for k in ("q", "qd")}
    obs, reward, done, info = env.step(a, reset_mask=m)
    assert bool(info["reset"][m].all()) and not bool(info["reset"][~m].any())
    for _ in range(6):
        obs, reward, done, info = env.step(a)
    add_b = env._courses[1][0]
    z = info["base"][:, 2]
    assert add_b > 1.0 and float((z[m] - (0.28 + add_b)).abs().max()) < 0.06          # on top of the down-stair course
    assert float((z[~m] - z_a[~m]).abs().max()) < 0.02                               # the others still stand where they stood
    yaw = info["pose"][:, 2]
    assert float((yaw[m] - 0.5).abs().max()) < 0.05 and float(yaw[~m].abs().max()) < 0.05
    x = info["base"][:, 0]
    assert float(x[m].min()) > -0.25 and float(x[m].max()) < 0.15 and float(x[m].std()) > 0.03      # U(-0.2, 0.1)
    assert not bool(done.any()) and torch.isfinite(obs).all()
    # a third course for a few of the others; a fourth does not fit three slots while all are in use
    m2 = torch.arange(n, device=DEV) < 4
    env.configure_reset(m2 & ~m, hardset=True, mode="slope", stepwidth=0.3, slope=0.3, stepheight=0.05, env_vec=[])
    assert sorted(set(env.terrain_id.cpu().tolist())) == [0, 1, 2]
    with pytest.raises(Exception, match="terrain slots"):
        env.configure_reset(torch.arange(n, device=DEV) == 7, hardset=True, mode="slope", stepwidth=0.3, slope=0.35, stepheight=0.05, env_vec=[])


def test_reference_etg_fixture_walks_the_a1_like_robot_on_stairstair():
    """The only quadrupedal artefact the reference ships, quadrupedal/ESStair_origin.npz (trained ETG weights), run the way
    quadrupedal/test_ETG.py:5 runs it — task "stairstair", ETG=1, zero policy action, 100 steps — on THIS engine with the repo's
    A1-shaped demo robot (examples/a1_like/a1_like.urdf; not pybullet_data's a1.urdf, which is absent). The Python side of that
    run is pinned to the reference in tests/golden/a1_env_episodes.npz (env_etg_fixture); here: behaviour, reported in DESIGN.md
    §3.7, nothing asserted against PyBullet — the gait moves the robot forward and no episode ends on a bad contact or a fall
    in the first 40 steps."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "a1_env_episodes.npz"))
    w, b = g["env_etg_fixture/w"], g["env_etg_fixture/b"]
    n = 256
    urdf = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "a1_like", "a1_like.urdf")
    env = metagym_amd.make("quadrupedal-v0", num_envs=n, urdf=urdf, device=DEV, task="stairstair", ETG=1, ETG_w=w, ETG_b=b)
    obs, info = env.reset()
    x0 = info["base"][:, 0].clone()
    a = torch.zeros(n, 12, dtype=torch.float64, device=DEV)
    first_done, bad_total = None, 0
    for k in range(100):
        obs, reward, done, info = env.step(a)
        bad_total += int(info["bad"].sum())
        if first_done is None and bool(done.any()):
            first_done = k
        if k == 39:
            x40, done40 = info["base"][:, 0].clone(), first_done
    x100 = info["base"][:, 0]
    print("ESStair_origin.npz on the a1-like robot, stairstair: forward progress %.3f m after 40 steps, %.3f m after 100; first done at step %s; "
          "%d bad contact points in 100 steps x %d robots" % (float((x40 - x0).mean()), float((x100 - x0).mean()), first_done, bad_total, n))
    assert torch.isfinite(obs).all()
    assert float((x40 - x0).mean()) > 0.0 and done40 is None


def test_quadrupedal_v0_runs_closed_loop_from_a_urdf():
    """`make("quadrupedal-v0", num_envs=..., urdf=...)` — no physics object: the robot file on the engine, 13 fused sub-steps
    per launch with the PD model inside. Zero actions hold (0, 0.9, -1.8) x 4: the robot stands on its four toes, nothing
    else touches, nothing terminates; on `slopestair` it stands on the start platform; an ETG gait moves it forward."""
    n = 512
    env = metagym_amd.make("quadrupedal-v0", num_envs=n, urdf=a1_like_urdf(), device=DEV)
    assert hasattr(env.physics, "fused_step")
    obs, info = env.reset()
    a = torch.zeros(n, 12, dtype=torch.float64, device=DEV)
    for _ in range(25):
        obs, reward, done, info = env.step(a)
        assert torch.isfinite(obs).all() and not bool(done.any())
    z = info["base"][:, 2]
    assert 0.2 < float(z.min()) and float(z.max()) < 0.3
    assert float(info["real_contact"].sum(dim=1).min()) >= 3.0 and int(info["bad"].max()) == 0
    assert float((env.robot.GetMotorAngles() - torch.as_tensor([0, 0.9, -1.8] * 4, device=DEV)).abs().max()) < 0.15
    # standing still the four toes carry the weight: sum of the normal forces = m g (12.621 kg x 10)
    total = env.physics.world()["foot_force"].sum(dim=1)
    assert float((total - 126.21).abs().max()) < 0.15 * 126.21
    # ... and the SimpleFootForceSensor stack (contact = 2) reports flags + force / 100 in its 8 entries
    env2 = metagym_amd.make("quadrupedal-v0", num_envs=n, urdf=a1_like_urdf(), device=DEV,
                            sensor_mode=dict(dis=1, motor=1, imu=1, contact=2, footpose=0))
    obs2, _ = env2.reset()
    for _ in range(25):
        obs2, _, _, info2 = env2.step(a)
    assert obs2.shape == (n, 41) and torch.equal(obs2[:, 3:7], info2["real_contact"])
    assert float((obs2[:, 7:11].sum(dim=1) * 100.0 - 126.21).abs().max()) < 0.15 * 126.21
    env = metagym_amd.make("quadrupedal-v0", num_envs=n, urdf=a1_like_urdf(), device=DEV, task="slopestair")
    obs, info = env.reset()
    for _ in range(15):
        obs, reward, done, info = env.step(a)
    assert float(info["base"][:, 2].min()) > 0.2 + env.add_height - 0.05 and not bool(done.any())
    # ETG open-loop trot with hand-set weights: the base advances
    w = np.tile([[0.03], [0.0], [0.02]], (1, 20)) * np.sin(np.linspace(0, 2 * np.pi, 20))
    env = metagym_amd.make("quadrupedal-v0", num_envs=n, urdf=a1_like_urdf(), device=DEV, ETG=1, ETG_w=w, ETG_b=np.zeros(3))
    obs, info = env.reset()
    x0 = info["base"][:, 0].clone()
    for _ in range(40):
        obs, reward, done, info = env.step(a)
    assert torch.isfinite(obs).all() and float((info["base"][:, 0] - x0).abs().mean()) > 0.01


def test_quadrupedal_without_a_robot_explains_itself():
    with pytest.raises(Exception, match="urdf="):
        metagym_amd.make("quadrupedal-v0", num_envs=4, device=DEV)


def test_body_damping_and_bullet_box_inertia_on_the_engine_match_the_oracle():
    """The two PyBullet-default behaviours DESIGN.md §3.4 lists as options, on the GPU (shape-generic wave kernel): humanoids
    with btMultiBody's 0.04 / 0.04 velocity damping and bounding-box inertias follow the oracle run with the same options
    (1e-9 over 10 env steps), and differ from the undamped run."""
    models = [variants.model("humanoid", preset="mujoco", inertia="bullet_box"),
              variants.model("humanoid", "TRAIN", 7, preset="mujoco", inertia="bullet_box")]
    n = 6
    env = MetaHumanoidEnv(num_envs=n, device=DEV, preset="mujoco", body_damping=(0.04, 0.04))
    plain = MetaHumanoidEnv(num_envs=n, device=DEV, preset="mujoco")
    rs = np.random.RandomState(1)
    noise = rs.uniform(-0.1, 0.1, (n, 17))
    oenvs = []
    for e_, mods in ((env, models), (plain, models)):
        e_.set_task(mods)
        e_.reset(joint_noise=noise)
    ids = env.task_id.cpu().numpy()
    for k in range(n):
        m = models[ids[k]]
        o = abd.WalkerEnv(m, prm=abd.Params(friction=0.8 * float(m.geom_friction), self_friction=float(m.geom_friction) ** 2,
                                            body_damping=(0.04, 0.04)))
        o.reset(noise[k])
        oenvs.append(o)
    worst = 0.0
    for t in range(10):
        a = rs.uniform(-1, 1, (n, 17)).astype(np.float32)
        env.step(torch.as_tensor(a))
        plain.step(torch.as_tensor(a))
        q, pos = env.q.cpu().numpy().T, env.pos.cpu().numpy().T
        for k in range(n):
            oenvs[k].step(a[k])
            worst = max(worst, np.abs(q[k] - oenvs[k].s.q).max(), np.abs(pos[k] - oenvs[k].s.pos).max())
    assert worst < 1e-9, worst
    assert float((env.q - plain.q).abs().max()) > 1e-4            # the option does something
    print("humanoid, body damping 0.04 + bullet_box inertia: max |state diff| GPU vs oracle over 10 steps %.2e" % worst)


def test_a1_gym_env_checkpoint_continues_bit_for_bit():
    """A1GymEnv.state_dict / load_state_dict (actuator history ring and counters, action filter, ETG output, sensor stack,
    reward bookkeeping, RNN frames, device clock, pending auto-reset mask, engine state): a fresh env loaded from a mid-episode
    checkpoint reproduces the original's next steps exactly, episodes ending (auto_reset) inside them."""
    n = 64
    w = np.tile([[0.03], [0.0], [0.02]], (1, 20)) * np.sin(np.linspace(0, 2 * np.pi, 20))
    mk = lambda: metagym_amd.make("quadrupedal-v0", num_envs=n, urdf=a1_like_urdf(), device=DEV, ETG=1, ETG_w=w, ETG_b=np.zeros(3),
                                  filter_=1, control_latency=0.0057, auto_reset=True,
                                  sensor_mode=dict(dis=1, motor=1, imu=1, contact=1, ETG=1, RNN=dict(time_steps=2, time_interval=1, mode="stack")))
    a = mk()
    a.reset()
    rs = np.random.RandomState(0)
    acts = [torch.as_tensor(rs.uniform(-0.5, 0.5, (n, 12)), device=DEV) for _ in range(10)]
    acts[2][::4] = 3.0                                      # every fourth robot is thrown off balance: episodes end
    for k in range(4):
        a.step(acts[k])
    sd = a.state_dict()
    outs = [a.step(acts[k]) for k in range(4, 10)]
    outs = [(o.clone(), r.clone(), d.clone()) for o, r, d, _ in outs]
    b = mk()
    b.reset()
    b.load_state_dict(sd)
    ends = 0
    for k in range(4, 10):
        o, r, d, _ = b.step(acts[k])
        assert torch.equal(o, outs[k - 4][0]) and torch.equal(r, outs[k - 4][1]) and torch.equal(d, outs[k - 4][2]), k
        ends += int(d.sum())
    assert ends > 0


@pytest.mark.parametrize("slots", [3, 1])
def test_a1_gym_env_checkpoint_carries_courses_headings_and_start_noise(slots):
    """what reset(**kwargs) / configure_reset() leave behind is part of the checkpoint — the terrain table and
    every robot's course (or the batch's one hardset course), the reward's per-course stretches, start headings, the x-noise flags
    and THEIR generator, new ETG parameters, the physics' reset pose. A fresh env (constructed on the task's default course, reset
    with defaults) loaded from a checkpoint taken after all of those were changed continues bit for bit through auto-resets that
    use them — and stands where the original stands, not on course 0 with the default pose."""
    n = 48
    w = np.tile([[0.03], [0.0], [0.02]], (1, 20)) * np.sin(np.linspace(0, 2 * np.pi, 20))
    mk = lambda: metagym_amd.make("quadrupedal-v0", num_envs=n, urdf=a1_like_urdf(), device=DEV, task="slopestair", terrain_slots=slots,
                                  ETG=1, ETG_w=w, ETG_b=np.zeros(3), auto_reset=True, seed=9)
    a = mk()
    a.reset()
    zero = torch.zeros(n, 12, dtype=torch.float64, device=DEV)
    course = dict(hardset=True, mode="downstair", stepwidth=0.28, slope=0.3, stepheight=0.06, env_vec=[])
    if slots > 1:
        m = torch.arange(n, device=DEV) % 2 == 1
        a.configure_reset(m, yaw=0.5, x_noise=True, ETG_w=0.5 * w, **course)
        a.step(zero, reset_mask=m)
    else:
        m = torch.ones(n, dtype=torch.bool, device=DEV)
        a.reset(yaw=0.5, x_noise=True, ETG_w=0.5 * w, **course)
    for _ in range(3):
        a.step(zero)
    sd = a.state_dict()
    rs = np.random.RandomState(3)
    acts = [torch.as_tensor(rs.uniform(-0.3, 0.3, (n, 12)), device=DEV) for _ in range(12)]
    acts[1][::3] = 3.0                                       # every third robot falls: auto-resets with ITS course, heading, noise
    acts[6][1::3] = 3.0
    outs = []
    for k in range(12):
        o, r, d, info = a.step(acts[k])
        outs.append((o.clone(), r.clone(), d.clone(), info["base"].clone(), info["pose"].clone()))
    b = mk()
    b.reset()
    b.load_state_dict(sd)
    assert b._first_reset is False and b._x_noise_any and np.array_equal(b.path.etg_w(), 0.5 * w)
    if slots > 1:
        assert torch.equal(b.terrain_id, a.terrain_id) and sorted(b._courses) == [0, 1] and b._courses[1][0] == a._courses[1][0]
        assert torch.equal(b.physics.env._terrain_t, a.physics.env._terrain_t)
    else:
        assert b.add_height == a.add_height and b.default_pose == a.default_pose and len(b.terrain_boxes) == len(a.terrain_boxes)
    ends = 0
    for k in range(12):
        o, r, d, info = b.step(acts[k])
        want = outs[k]
        assert torch.equal(o, want[0]) and torch.equal(r, want[1]) and torch.equal(d, want[2]), k
        assert torch.equal(info["base"], want[3]) and torch.equal(info["pose"], want[4]), k
        ends += int(d[m].sum())
    assert ends > 0                                          # robots of the changed course restarted inside the compared steps ...
    add_b = a._courses[1][0] if slots > 1 else a.add_height
    z, yaw = info["base"][:, 2], info["pose"][:, 2]
    assert add_b > 1.0 and float(z[m].min()) > add_b          # ... on the down-stair course's platform, not on course 0
    fresh = m & (b._substeps_dev < 13 * 8)                   # those that restarted within the last 8 steps still face their heading
    assert int(fresh.sum()) > 0 and float((yaw[fresh] - 0.5).abs().max()) < 0.2


@pytest.mark.parametrize("mode", ["position_env_gains", "hybrid", "torque"])
def test_fused_actuation_modes_equal_the_substep_path_bit_for_bit(mode):
    """The engine's in-launch actuators beyond the A1 default: POSITION with per-robot gains (what
    locomotion_gym_env.py:388-392 draws), HYBRID (laikago_motor.py:143-153) and TORQUE (:125-128) — 13 fused sub-steps against
    the same loop run sub-step by sub-step through mg_a1_apply_action (whose motor model is pinned to the reference, all three
    modes): torques, observation history and engine state identical."""
    from metagym_amd.quadrupedal import A1Actuators, MotorControlMode
    n = 48
    rs = np.random.RandomState(5)
    cmode = {"position_env_gains": MotorControlMode.POSITION, "hybrid": MotorControlMode.HYBRID, "torque": MotorControlMode.TORQUE}[mode]
    runs = []
    for fused in (True, False):
        phys = A1Physics(n, urdf=a1_like_urdf(), device=DEV, fused=fused)
        act = A1Actuators(n, DEV, motor_control_mode=cmode, control_latency=0.0041)
        if mode == "position_env_gains":
            act.SetMotorGains(torch.as_tensor(np.random.RandomState(1).uniform(60, 110, (n, 12)), device=DEV),
                              torch.as_tensor(np.random.RandomState(2).uniform(1.0, 4.0, (n, 12)), device=DEV))
        act.SetMotorStrengthRatios(np.linspace(0.8, 1.1, 12))
        act.Reset()
        act.ReceiveObservation(*phys.reset(None))
        runs.append((phys, act))
    target = np.array([0, 0.9, -1.8] * 4)
    for k in range(6):
        if mode == "torque":
            cmd = rs.uniform(-8, 8, (n, 12))
        elif mode == "hybrid":
            cmd = np.zeros((n, 12, 5))
            cmd[..., 0] = target + rs.uniform(-0.2, 0.2, (n, 12)); cmd[..., 1] = rs.uniform(50, 100, (n, 12))
            cmd[..., 2] = rs.uniform(-0.5, 0.5, (n, 12)); cmd[..., 3] = rs.uniform(1, 3, (n, 12)); cmd[..., 4] = rs.uniform(-2, 2, (n, 12))
            cmd = cmd.reshape(n, 60)
        else:
            cmd = target + rs.uniform(-0.2, 0.2, (n, 12))
        a = torch.as_tensor(cmd, device=DEV)
        (pf, af), (pu, au) = runs
        tf = af.StepFused(a, pf.fused_step)
        tu = au.Step(a, pu.substep)
        assert torch.equal(tf, tu), (mode, k)
        assert torch.equal(af.GetControlObservation(), au.GetControlObservation())
        for key in ("pos", "rot", "vel", "omega", "q", "qd"):
            assert torch.equal(getattr(pf.env, key), getattr(pu.env, key)), (mode, k, key)
    assert torch.isfinite(tf).all() and float(tf.abs().max()) > 1.0


def test_external_push_on_the_engine_matches_the_oracle_and_newton():
    """mg_walker_params.ext_wrench (RandomWrapper's applyExternalForce on the base, LINK_FRAME): acts in the first sub-step of a
    launch only. One sub-step against the oracle (1e-9), and in free flight the robot's momentum changes by R f dt, once."""
    n = 16
    phys = A1Physics(n, urdf=a1_like_urdf(), device=DEV, foot_links=A1_LIKE_TOES, gravity=0.0)
    m = phys.model
    phys.reset(None)
    e = phys.env
    e.pos[2] += 5.0                                              # free flight
    rs = np.random.RandomState(3)
    e.vel.copy_(torch.as_tensor(rs.uniform(-0.5, 0.5, (3, n))))
    e.omega.copy_(torch.as_tensor(rs.uniform(-1, 1, (3, n))))
    e.qd.copy_(torch.as_tensor(rs.uniform(-1, 1, (12, n))))
    force, pos = rs.uniform(-40, 40, (n, 3)), rs.uniform(-0.2, 0.2, (n, 3))
    keys = ("pos", "rot", "vel", "omega", "q", "qd")
    st = {k: getattr(e, k).cpu().numpy() for k in keys}
    prm = abd.Params(contact_margin=MARGINS(m), dt=0.002, substeps=1, iterations=23, erp=0.2, friction=5.0, sphere_friction=m.sph_friction, self_collision=False, gravity=0.0)
    phys.apply_external_force(torch.as_tensor(force), torch.as_tensor(pos))
    tau = np.zeros((12, n))
    log = torch.empty(3, 43, n, dtype=torch.float64, device=DEV)
    e.step_actuated(torch.as_tensor(tau, device=DEV), raw_torque=True, n_substeps=3, log=log)        # 3 sub-steps, pushed in the first
    worst = 0.0
    for k in range(n):
        s = abd.State(m)
        s.pos, s.rot, s.v, s.w = st["pos"][:, k].copy(), st["rot"][:, k].reshape(3, 3).copy(), st["vel"][:, k].copy(), st["omega"][:, k].copy()
        s.q, s.qd = st["q"][:, k].copy(), st["qd"][:, k].copy()
        P0, _ = abd.momentum(m, s)
        R0 = s.rot.copy()
        abd.substep(m, s, tau[:, k], prm, ext=(force[k], pos[k] + m.root_inertial_pos))
        P1, _ = abd.momentum(m, s)
        assert np.allclose(P1 - P0, R0 @ force[k] * 0.002, rtol=0, atol=5e-4)           # impulse = F dt (up to 0.08 N s; the rest is
        #                                                                                  the integrator's O(dt) drift of a spinning robot)
        abd.substep(m, s, tau[:, k], prm)
        abd.substep(m, s, tau[:, k], prm)
        P3, _ = abd.momentum(m, s)
        assert np.allclose(P3, P1, rtol=0, atol=1e-3)                                    # ... and only once
        d = max(np.abs(e.q.cpu().numpy()[:, k] - s.q).max(), np.abs(e.pos.cpu().numpy()[:, k] - s.pos).max(),
                np.abs(e.vel.cpu().numpy()[:, k] - s.v).max(), np.abs(e.omega.cpu().numpy()[:, k] - s.w).max())
        worst = max(worst, d)
    assert worst < 1e-9, worst
    assert float(phys._ext.abs().max()) > 0                     # (step_actuated alone does not clear it; A1Physics.substep / fused_step do)


def test_random_force_and_dynamic_param_closed_loop_on_the_engine():
    """RandomWrapper's pushes (`random_param`) and explicit dynamics (`dynamic_param`) on the URDF robot: 20-50 N pushes every env
    step for 50 of every 100, observation entries force_vec / dynamic_vec; the robots keep standing, the pushed ones drift."""
    n = 256
    mode = dict(dis=1, motor=1, imu=1, contact=1, footpose=0, force_vec=1, dynamic_vec=1)
    env = metagym_amd.make("quadrupedal-v0", num_envs=n, urdf=a1_like_urdf(), device=DEV, sensor_mode=mode, seed=3,
                           random_param={"random_force": 1}, dynamic_param={"control_latency": 17.0, "footfriction": 1.7, "basemass": 1.1})
    quiet = metagym_amd.make("quadrupedal-v0", num_envs=n, urdf=a1_like_urdf(), device=DEV)
    assert abs(env.physics.base_mass - 1.1 * 4.7) < 1e-12 and env.robot._cfg.control_latency == pytest.approx(0.017)
    obs, info = env.reset()
    quiet.reset()
    assert obs.shape == (n, 37 + 6 + 3)
    assert torch.allclose(obs[:, 43:46], torch.tensor([0.017, 1.7, 1.1 * 4.7], dtype=torch.float64, device=DEV).expand(n, 3))
    a = torch.zeros(n, 12, dtype=torch.float64, device=DEV)
    on_steps = 0
    for k in range(60):
        obs, reward, done, info = env.step(a)
        quiet.step(a)
        fv = obs[:, 37:43]
        assert torch.equal(fv, info["force_vec"])
        on = bool((fv.abs().sum(dim=1) > 0).all())
        assert on == (k + 2 < 50)                               # counter = k + 2 after this step (the hidden step counted once)
        on_steps += on
        if on:
            mag = (fv[:, 3:6] * 50.0).norm(dim=1)
            assert float(mag.min()) >= 20.0 - 1e-9 and float(mag.max()) <= 50.0 + 1e-9
    assert on_steps == 48 and torch.isfinite(obs).all()
    drift = (env.physics.world()["base"][:, :2] - quiet.physics.world()["base"][:, :2]).norm(dim=1)
    assert float(drift.mean()) > 1e-4 and float(done.double().mean()) < 0.2      # (a 35 N push for 2 ms is a 0.07 N s nudge per env step)


def test_captured_step_with_pushes_keeps_drawing_forces():
    """capture_step() with random_force on: the pushes' device generator is registered with the hipGraph, so replays keep
    advancing it — a new force appears every 100 env steps, and the step stays finite."""
    n = 64
    mode = dict(dis=1, motor=1, imu=1, contact=1, footpose=0, force_vec=1)
    env = metagym_amd.make("quadrupedal-v0", num_envs=n, urdf=a1_like_urdf(), device=DEV, sensor_mode=mode, seed=1,
                           random_param={"random_force": 1})
    env.reset()
    replay = env.capture_step()
    env.reset()
    a = torch.zeros(n, 12, dtype=torch.float64, device=DEV)
    forces = []
    for k in range(205):
        obs, reward, done, info = replay(a)
        forces.append(obs[:, 37:43].clone())
    assert torch.isfinite(obs).all()
    on = torch.stack([(f.abs().sum(dim=1) > 0).all() for f in forces]).cpu().numpy()
    c = np.arange(205) + 2                                              # counter after replay k (reset's hidden step counted once)
    assert np.array_equal(on, (c % 100) < 50)
    first, second, third = forces[0], forces[100], forces[199]          # counters 2, 102, 201: three different draws
    assert not torch.equal(first, second) and not torch.equal(second, third)
