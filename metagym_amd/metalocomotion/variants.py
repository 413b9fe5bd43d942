"""The MetaLocomotion task distribution without the reference's asset directory.

The reference ships its tasks as 2 x 385 MJCF files (metalocomotion/envs/assets/humanoids/, ants/): one base
robot plus 384 body variants written by `gen_variant_humanoids.py` / `gen_variant_ants.py`
(assets/humanoids/gen_variant_humanoids.py:33-87, assets/ants/gen_variant_ants.py:33-67) from *unseeded*
random "patterns" — 3 numbers per humanoid (pelvis width, thigh length, shin length factors), 12 per ant
(three segment factors per leg) — split into the 64 farthest from the mean (`ood`), 256 `tra` and 64 `tst`
(`pattern_ood_clustering`, :16-31). This module restates that generator:

  * the two base robots as compact specs (numbers from assets/humanoids/humanoid.xml:15-83 and
    assets/ants/ant.xml:11-62; only what the physics reads: bodies, hinge joints, capsule / sphere geoms, defaults);
  * `apply_pattern`: the edits `reconfig_xml` makes, in the same float64 arithmetic;
  * `variant_patterns.npz`: the 384 patterns per robot, recovered from the shipped files by
    oracle/extract_variant_patterns.py (the generator was never seeded, so the shipped files ARE the
    distribution) such that every regenerated model equals the parse of the shipped file number for number
    (tests/test_walker_variants.py; digests of all 768 committed for machines without the reference).

`sample_task('TRAIN')` of the envs therefore works on a box that has no reference tree, with the reference's
file names (`humanoid_var_tra_017.xml`); `models('humanoid', 'TRAIN')` is BASELINE config C4's input
(`humanoid_var_tra_000...255.xml` round-robin).
"""
import os
import re
import xml.etree.ElementTree as ET

import numpy as np

from .mjcf import load_mjcf, preset_options

SPLITS = {"TRAIN": "tra", "TEST": "tst", "OOD": "ood"}
FEET = {"humanoid": ("right_foot", "left_foot"),
        "ant": ("front_left_foot", "front_right_foot", "left_back_foot", "right_back_foot")}
_PATTERNS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variant_patterns.npz")


# ---- compact robot specs -----------------------------------------------------------------------------
def _body(name, pos, *items, quat=None):
    return {"name": name, "pos": [float(v) for v in pos], "quat": quat, "items": list(items)}


def _capsule(name, fromto, radius):
    return {"geom": "capsule", "name": name, "fromto": [float(v) for v in fromto], "size": float(radius)}


def _sphere(name, pos, radius):
    return {"geom": "sphere", "name": name, "pos": [float(v) for v in pos], "size": float(radius)}


def _hinge(name, axis, lo, hi, pos=(0, 0, 0), **extra):
    return {"joint": name, "axis": [float(v) for v in axis], "range": [float(lo), float(hi)],
            "pos": [float(v) for v in pos], "extra": {k: float(v) for k, v in extra.items()}}


def _humanoid_leg(side, y, ty):
    """humanoid.xml:29-58; the left leg mirrors the right one in y (and in the hip x / z axes). ty: the thigh's
    slight outward lean (y of its capsule's far end = y of the shin anchor)."""
    s = 1.0 if side == "right" else -1.0
    hip = dict(damping=5, armature=0.01)
    return _body(side + "_thigh", (0, y, -0.04),
                 _hinge(side + "_hip_x", (s, 0, 0), -25, 5, stiffness=10, **hip),
                 _hinge(side + "_hip_z", (0, 0, s), -60, 35, stiffness=10, **hip),
                 _hinge(side + "_hip_y", (0, 1, 0), -120, 20, stiffness=20, **hip),
                 _capsule(side + "_thigh1", (0, 0, 0, 0, ty, -0.34), 0.06),
                 _body(side + "_shin", (0, ty, -0.403),
                       _hinge(side + "_knee", (0, -1, 0), -160, -2, pos=(0, 0, 0.02), stiffness=1, armature=0.0060),
                       _capsule(side + "_shin1", (0, 0, 0, 0, 0, -0.3), 0.049),
                       _body(side + "_foot", (0, 0, -0.45), _sphere(side + "_foot", (0, 0, 0.1), 0.075))))


def _humanoid_arm(side, y):
    """humanoid.xml:60-81"""
    r = side == "right"
    return _body(side + "_upper_arm", (0, y, 0.06),
                 _hinge(side + "_shoulder1", (2, 1 if r else -1, 1), *((-85, 60) if r else (-60, 85)), stiffness=1, armature=0.0068),
                 _hinge(side + "_shoulder2", (0, -1 if r else 1, 1), *((-85, 60) if r else (-60, 85)), stiffness=1, armature=0.0051),
                 _capsule(side + "_uarm1", (0, 0, 0, 0.16, y / abs(y) * 0.16, -0.16), 0.04),
                 _body(side + "_lower_arm", (0.18, y / abs(y) * 0.18, -0.18),
                       _hinge(side + "_elbow", (0, -1, 1 if r else -1), -90, 50, stiffness=0, armature=0.0028),
                       _capsule(side + "_larm", (0.01, -y / abs(y) * 0.01, 0.01, 0.17, -y / abs(y) * 0.17, 0.17), 0.031),
                       _sphere(side + "_hand", (0.18, -y / abs(y) * 0.18, 0.18), 0.04)))


def humanoid_spec():
    """assets/humanoids/humanoid.xml: 13 bodies, 17 hinges; joint defaults armature 1, damping 1 (:4)."""
    waist = dict(damping=5, armature=0.02)
    tilt = (1.0, 0.0, -0.002, 0.0)
    torso = _body("torso", (0, 0, 1.4),
                  _capsule("torso1", (0, -0.07, 0, 0, 0.07, 0), 0.07),
                  _sphere("head", (0, 0, 0.19), 0.09),
                  _capsule("uwaist", (-0.01, -0.06, -0.12, -0.01, 0.06, -0.12), 0.06),
                  _body("lwaist", (-0.01, 0, -0.260),
                        _capsule("lwaist", (0, -0.06, 0, 0, 0.06, 0), 0.06),
                        _hinge("abdomen_z", (0, 0, 1), -45, 45, pos=(0, 0, 0.065), stiffness=20, **waist),
                        _hinge("abdomen_y", (0, 1, 0), -75, 30, pos=(0, 0, 0.065), stiffness=10, **waist),
                        _body("pelvis", (0, 0, -0.165),
                              _hinge("abdomen_x", (1, 0, 0), -35, 35, pos=(0, 0, 0.1), stiffness=10, **waist),
                              _capsule("butt", (-0.02, -0.07, 0, -0.02, 0.07, 0), 0.09),
                              _humanoid_leg("right", -0.1, 0.01), _humanoid_leg("left", 0.1, -0.01), quat=tilt),
                        quat=tilt),
                  _humanoid_arm("right", -0.17), _humanoid_arm("left", 0.17))
    return {"model": "humanoid", "joint_default": {"armature": 1.0, "damping": 1.0},
            "geom_default": {"friction": "0.8 0.1 0.1"}, "root": torso}


def ant_spec():
    """assets/ants/ant.xml: torso sphere + 4 legs of (fixed hip stub, hip hinge, ankle hinge); geom density 5,
    friction 1.5 (:9)."""
    legs = [("front_left_leg", "aux_1", "front_left_foot", 1, 1, "hip_1", "ankle_1", (-1, 1, 0), (30, 100),
             ("aux_1_geom", "left_leg_geom", "left_ankle_geom")),
            ("front_right_leg", "aux_2", "front_right_foot", -1, 1, "hip_2", "ankle_2", (1, 1, 0), (-100, -30),
             ("aux_2_geom", "right_leg_geom", "right_ankle_geom")),
            ("left_back_leg", "aux_3", "left_back_foot", -1, -1, "hip_3", "ankle_3", (-1, 1, 0), (-100, -30),
             ("aux_3_geom", "back_leg_geom", "third_ankle_geom")),
            ("right_back_leg", "aux_4", "right_back_foot", 1, -1, "hip_4", "ankle_4", (1, 1, 0), (30, 100),
             ("aux_4_geom", "rightback_leg_geom", "fourth_ankle_geom"))]
    items = [_sphere("torso_geom", (0, 0, 0), 0.25)]
    for leg, aux, foot, sx, sy, hip, ankle, ankle_axis, ankle_range, gn in legs:
        a, b = 0.2 * sx, 0.2 * sy
        items.append(_body(leg, (0, 0, 0),
                           _capsule(gn[0], (0, 0, 0, a, b, 0), 0.08),
                           _body(aux, (a, b, 0),
                                 _hinge(hip, (0, 0, 1), -40, 40),
                                 _capsule(gn[1], (0, 0, 0, a, b, 0), 0.08),
                                 _body(foot, (a, b, 0),
                                       _hinge(ankle, ankle_axis, *ankle_range),
                                       _capsule(gn[2], (0, 0, 0, 2 * a, 2 * b, 0), 0.08)))))
    return {"model": "ant", "joint_default": {"armature": 1.0, "damping": 1.0},
            "geom_default": {"friction": "1.5 0.1 0.1", "density": "5.0"}, "root": _body("torso", (0, 0, 0.75), *items)}


def base_spec(robot):
    if robot == "humanoid":
        return humanoid_spec()
    if robot == "ant":
        return ant_spec()
    raise ValueError("unknown robot %r (humanoid, ant)" % (robot,))


# ---- the variant generator ---------------------------------------------------------------------------
def _child_bodies(body):
    return [it for it in body["items"] if "items" in it]


def _first_capsule(body):
    return next(it for it in body["items"] if it.get("geom") == "capsule")


def _scale(values, factor):
    return [v * factor for v in values]


def apply_pattern(robot, spec, pattern):
    """`reconfig_xml` of the reference's generators on the spec, same float64 products (in place; returns spec).

    humanoid (gen_variant_humanoids.py:33-76), pattern = (pelvis width, thigh, shin factors):
      pelvis capsule y extent x p0; both thigh anchors' y x p0; thigh capsule and shin anchor x p1; shin capsule
      and foot anchor x p2; torso height += 0.403 (p1 - 1) + 0.45 (p2 - 1) - 0.20.
    ant (gen_variant_ants.py:33-56), pattern = 3 factors per leg in document order:
      each of the leg's three capsules x its factor; the next body is re-anchored at the scaled capsule's far end."""
    p = [float(v) for v in pattern]
    root = spec["root"]
    if robot == "humanoid":
        assert len(p) == 3
        pelvis = _child_bodies(_child_bodies(root)[0])[0]
        butt = _first_capsule(pelvis)
        butt["fromto"][1] *= p[0]
        butt["fromto"][4] *= p[0]
        root["pos"][2] += 0.403 * (p[1] - 1.0) + 0.45 * (p[2] - 1.0) - 0.20
        for thigh in _child_bodies(pelvis):
            shin = _child_bodies(thigh)[0]
            foot = _child_bodies(shin)[0]
            thigh["pos"][1] *= p[0]
            _first_capsule(thigh)["fromto"] = _scale(_first_capsule(thigh)["fromto"], p[1])
            shin["pos"] = _scale(shin["pos"], p[1])
            _first_capsule(shin)["fromto"] = _scale(_first_capsule(shin)["fromto"], p[2])
            foot["pos"] = _scale(foot["pos"], p[2])
    elif robot == "ant":
        assert len(p) == 12
        for k, leg in enumerate(_child_bodies(root)):
            aux = _child_bodies(leg)[0]
            foot = _child_bodies(aux)[0]
            for seg, f, nxt in ((leg, p[3 * k], aux), (aux, p[3 * k + 1], foot), (foot, p[3 * k + 2], None)):
                cap = _first_capsule(seg)
                cap["fromto"] = _scale(cap["fromto"], f)
                if nxt is not None:
                    nxt["pos"] = list(cap["fromto"][3:])
    else:
        raise ValueError(robot)
    return spec


def _fmt(values):
    return " ".join(repr(float(v)) for v in values)


def _emit(parent, body):
    e = ET.SubElement(parent, "body", name=body["name"], pos=_fmt(body["pos"]))
    if body["quat"] is not None:
        e.set("quat", _fmt(body["quat"]))
    for it in body["items"]:
        if "items" in it:
            _emit(e, it)
        elif "joint" in it:
            j = ET.SubElement(e, "joint", name=it["joint"], type="hinge", axis=_fmt(it["axis"]), pos=_fmt(it["pos"]),
                              range=_fmt(it["range"]))
            for k, v in it["extra"].items():
                j.set(k, repr(v))
        elif it["geom"] == "capsule":
            ET.SubElement(e, "geom", name=it["name"], type="capsule", fromto=_fmt(it["fromto"]), size=repr(it["size"]))
        else:
            ET.SubElement(e, "geom", name=it["name"], type="sphere", pos=_fmt(it["pos"]), size=repr(it["size"]))


def mjcf_text(robot, pattern=None):
    """MJCF (the subset the reference's assets use) of the base robot, or of the variant with `pattern`."""
    spec = base_spec(robot)
    if pattern is not None:
        apply_pattern(robot, spec, pattern)
    root = ET.Element("mujoco", model=spec["model"])
    ET.SubElement(root, "compiler", angle="degree", inertiafromgeom="true")
    d = ET.SubElement(root, "default")
    ET.SubElement(d, "joint", limited="true", **{k: repr(v) for k, v in spec["joint_default"].items()})
    ET.SubElement(d, "geom", condim="3", **spec["geom_default"])
    _emit(ET.SubElement(root, "worldbody"), spec["root"])
    return ET.tostring(root, encoding="unicode")


# ---- the shipped distribution ------------------------------------------------------------------------
_cache = {}


def patterns(robot, split):
    """[n_variants, 3 | 12] float64: the patterns of `<robot>_var_<tra|tst|ood>_000...` in index order."""
    tag = SPLITS.get(split, split)
    if "npz" not in _cache:
        if not os.path.exists(_PATTERNS):
            raise FileNotFoundError("%s is missing (written by oracle/extract_variant_patterns.py)" % _PATTERNS)
        with np.load(_PATTERNS) as z:
            _cache["npz"] = {k: z[k] for k in z.files}
    return _cache["npz"]["%s_%s" % (robot, tag)]


def task_names(robot, split):
    """The reference's file names for a split ('TRAIN' | 'TEST' | 'OOD'), sorted."""
    tag = SPLITS.get(split, split)
    return ["%s_var_%s_%03d.xml" % (robot, tag, i) for i in range(len(patterns(robot, tag)))]


def model(robot, split=None, index=0, preset=None, **options):
    """Parsed `Model` of the base robot (`split=None`) or of variant `index` of a split. `preset` ('bullet', the default, or
    'mujoco') and `options` (inertia / com / armature / stiffness): the loader's reading of the file, mjcf.load_mjcf."""
    opt = preset_options(preset, **options)
    key = (robot, SPLITS.get(split, split), int(index) if split is not None else -1) + tuple(sorted((k, str(v)) for k, v in opt.items()))
    if key not in _cache:
        pat = None if split is None else patterns(robot, split)[int(index)]
        _cache[key] = load_mjcf(mjcf_text(robot, pat), foot_names=FEET[robot], preset=preset, **options)
    return _cache[key]


def models(robot, split, preset=None, **options):
    return [model(robot, split, i, preset=preset, **options) for i in range(len(patterns(robot, split)))]


_NAME = re.compile(r"^(humanoid|ant)(?:_var_(tra|tst|ood)_(\d{3}))?\.xml$")


def model_from_task_name(name, preset=None):
    """`humanoid_var_tra_017.xml` / `ant.xml` -> Model, or None if `name` is not one of the reference's file names."""
    m = _NAME.match(os.path.basename(name))
    if not m:
        return None
    robot, tag, idx = m.groups()
    if tag is None:
        return model(robot, preset=preset)
    if int(idx) >= len(patterns(robot, tag)):
        return None
    return model(robot, tag, int(idx), preset=preset)
