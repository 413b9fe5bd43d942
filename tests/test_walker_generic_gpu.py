"""Shape-generic instantiations of the walker engine: robots that are neither the humanoid, the ant nor the A1 — synthetic
"centipedes" with 4 / 8 / 10 / 12 / 14 / 17 / 20 hinges (10 ... 26 generalized coordinates, i.e. every register-slot count
14 / 18 / 23 / 30 both with spare slots and with the joint count equal to the slot count) — follow the numpy oracle
(oracle/abd.py) to float64 round-off. Bodies without joints, chains of different lengths, up to 6 legs. GPU box only."""
import xml.etree.ElementTree as ET

import numpy as np
import pytest
import torch

from oracle import abd

pytestmark = pytest.mark.gpu


def _centipede(legs):
    """Torso sphere + one leg per entry of `legs`: a jointless stub body, then a chain of capsule segments, segment s with
    legs[i][s] hinges on it (1-3, different axes through the segment's origin), the last segment being the foot."""
    from metagym_amd.metalocomotion.variants import _body, _capsule, _emit, _hinge, _sphere
    items = [_sphere("torso_geom", (0, 0, 0), 0.25)]
    feet = []
    nl = len(legs)
    for i, segs in enumerate(legs):
        ang = 2.0 * np.pi * (i + 0.5) / nl
        a, b = 0.22 * np.cos(ang), 0.22 * np.sin(ang)
        axes = [((0, 0, 1), (-35, 35)), ((-b, a, 0), (10, 70)), ((a, b, 0.3), (-30, 30))]
        chain = None
        for s in reversed(range(len(segs))):        # build the chain from the foot inwards
            name = "leg%d_seg%d" % (i, s) if s < len(segs) - 1 else "foot%d" % i
            seg = [_hinge("j%d_%d_%d" % (i, s, h), axes[(h + s) % 3][0], *axes[(h + s) % 3][1]) for h in range(segs[s])]
            last = s == len(segs) - 1
            seg.append(_capsule("g%d_%d" % (i, s), (0, 0, 0, 0.5 * a if last else a, 0.5 * b if last else b, -0.32 if last else 0.0), 0.07))
            if chain is not None:
                seg.append(chain)
            chain = _body(name, (a, b, 0.0), *seg)
        items.append(_body("stub%d" % i, (0, 0, 0), _capsule("stub_geom%d" % i, (0, 0, 0, a, b, 0), 0.07), chain))
        feet.append("foot%d" % i)
    root = ET.Element("mujoco", model="centipede")
    ET.SubElement(root, "compiler", angle="degree", inertiafromgeom="true")
    d = ET.SubElement(root, "default")
    ET.SubElement(d, "joint", limited="true", armature="0.5", damping="0.8")
    ET.SubElement(d, "geom", condim="3", friction="1.2 0.1 0.1", density="8.0")
    _emit(ET.SubElement(root, "worldbody"), _body("torso", (0, 0, 0.6), *items))
    return ET.tostring(root, encoding="unicode"), tuple(feet)


# hinges per segment per leg -> (generalized coordinates, kernel the launch picks)
ROBOTS = {"4 hinges (10 dof, <14, any>)": [[1], [1], [1], [1]],
          "8 hinges (14 dof, <14, 8 joints>)": [[2]] * 4,
          "10 hinges (16 dof, <18, any>)": [[2, 1], [1, 1], [2, 1], [1, 1]],
          "12 hinges (18 dof, <18, 12 joints>)": [[2, 1]] * 4,
          "14 hinges (20 dof, <23, any>)": [[3, 1], [2, 1], [3, 1], [2, 1]],
          "17 hinges (23 dof, <23, 17 joints>)": [[3, 2], [2, 2], [2, 2], [2, 2]],
          "20 hinges (26 dof, <30, any>)": [[3, 2]] * 4,
          "24 hinges (30 dof, every slot of <30, any>)": [[3, 3]] * 4,
          "snake: one 14-segment chain (29 hops deep: 5 + 4 scan rounds)": [[1] * 14],
          "stiff: 4 one-hinge legs, joint chains one joint deep": [[1]] * 4}


@pytest.mark.parametrize("preset", ["bullet", "mujoco"])
@pytest.mark.parametrize("name", list(ROBOTS) + ["self-collision: " + k for k in list(ROBOTS)[3:5]] +
                         ["lane mapping: " + k for k in (list(ROBOTS)[0], list(ROBOTS)[4])])
def test_generic_robot_follows_the_oracle(name, preset):
    """Both readings of the MJCF text (mjcf.PRESETS). "mujoco": joints with rotor inertia, damping and springs — GPU and oracle
    stay within 1e-8 over the 30 free-running env steps. "bullet" (the envs' default): none of the three, light links, body
    damping and the velocity clamp instead — undamped chains whipping at the clamp amplify float64 round-off like any chaotic
    system (3e-8 after 6 env steps on the 30-dof robot, > 1e-6 after 25; the 14-segment snake sits at the +-100 rad/s clamp
    throughout, where the two CPU restatements of the engine already differ by 2e-10 per SUB-step), which says nothing about the
    kernel: the engine's state is reloaded from the oracle's before every env step, so each step's 4 physics sub-steps are held
    to 1e-8 on their own."""
    import metagym_amd.metalocomotion as ml
    from metagym_amd.metalocomotion.mjcf import load_mjcf
    selfc, lane_map = name.startswith("self-collision: "), name.startswith("lane mapping: ")
    legs = ROBOTS[name.split(": ", 1)[1] if (selfc or lane_map) else name]
    text, feet = _centipede(legs)
    from walker_fixtures import world_kw
    m = load_mjcf(text, foot_names=feet, preset=preset)
    resync = 0 if preset == "mujoco" else 1
    nj = sum(sum(l) for l in legs)
    assert len(m.joint_lo) == nj and len(m.body_parent) == 1 + len(legs) + sum(len(l) for l in legs)

    class Centipede(ml.WalkerBatchEnv):
        robot_dir = None
        foot_list = feet
        power = 0.4
        motor_power = None
        alive_z = 0.15
        alive_bonus = 1.0
        initial_z = None

    n = 6
    kw = {"mapping": "lane"} if lane_map else {}
    env = Centipede(num_envs=n, device="cuda:0", max_steps=1000, self_collision=selfc, preset=preset, **kw)
    env.set_task([m])
    rs = np.random.RandomState(len(legs) * 100 + nj)
    noise = rs.uniform(-0.1, 0.1, (n, nj))
    env.reset(joint_noise=noise)
    oenvs = []
    for e in range(n):
        o = abd.WalkerEnv(m, prm=abd.Params(friction=0.8 * float(m.geom_friction), power=0.4, self_collision=selfc,
                                             self_friction=float(m.geom_friction) ** 2, **world_kw(m)),
                          motor_power=np.full(nj, 100.0), alive_z=0.15, alive_bonus=1.0, initial_z=None, torque_f32=False,
                          max_steps=1000)
        o.reset(noise[e])
        oenvs.append(o)
    worst, touched = 0.0, 0
    for t in range(30):
        if resync and t and t % resync == 0:
            col = lambda f: torch.as_tensor(np.stack([np.asarray(f(o), np.float64).reshape(-1) for o in oenvs], 1))
            env.load_state_dict(dict(pos=col(lambda o: o.s.pos), rot=col(lambda o: o.s.rot), vel=col(lambda o: o.s.v),
                                     omega=col(lambda o: o.s.w), q=col(lambda o: o.s.q), qd=col(lambda o: o.s.qd),
                                     potential=torch.as_tensor(np.asarray([o.potential for o in oenvs], np.float64)),
                                     feet_contact=col(lambda o: o.feet_contact).float(),
                                     steps=torch.as_tensor(np.asarray([o.steps for o in oenvs], np.int32))))
        a = rs.uniform(-0.7, 0.7, (n, nj)).astype(np.float32)
        env.step(torch.as_tensor(a))
        q, qd, pos = env.q.cpu().numpy().T, env.qd.cpu().numpy().T, env.pos.cpu().numpy().T
        fc = env.feet_contact.cpu().numpy().T
        for e in range(n):
            oenvs[e].step(a[e])
            s = oenvs[e].s
            worst = max(worst, np.abs(q[e] - s.q).max(), np.abs(pos[e] - s.pos).max(), 0.01 * np.abs(qd[e] - s.qd).max())
            assert np.allclose(q[e], s.q, rtol=0, atol=1e-8), (t, e, np.abs(q[e] - s.q).max())
            assert np.allclose(pos[e], s.pos, rtol=0, atol=1e-8), (t, e)
            assert np.array_equal(fc[e], oenvs[e].feet_contact), (t, e)
            touched += int(fc[e].sum())
    assert touched > 0            # somebody stood on a foot
    print("%s [%s]: max |state diff| GPU vs oracle over 30 env steps %.2e" % (name, preset, worst))
