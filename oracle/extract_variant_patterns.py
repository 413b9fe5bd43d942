#!/usr/bin/env python3
"""Recover the generator patterns of the reference's 2 x 384 shipped MetaLocomotion variants.

TEST / BUILD INFRASTRUCTURE; runs only in the build container (needs /root/reference):

    python oracle/extract_variant_patterns.py

`gen_variant_humanoids.py` / `gen_variant_ants.py` (metalocomotion/envs/assets/*/) drew their patterns from an
unseeded numpy RNG, so the shipped XML files are the only record of the task distribution. For every file this
script finds the float64 pattern p such that the product's generator (metagym_amd/metalocomotion/variants.py:
base spec + apply_pattern) reproduces EVERY number the reference generator wrote (each written number is
base * p[k] rounded once, so p[k] is searched among the few doubles around written / base that reproduce all
numbers depending on it; the torso height additionally ties p[1] and p[2] together), then verifies that the
regenerated Model equals the parse of the shipped file array for array.

Writes
  metagym_amd/metalocomotion/variant_patterns.npz   product data: humanoid_{tra,tst,ood} [256|64|64, 3],
                                                     ant_{tra,tst,ood} [.., 12]
  tests/golden/walker_variant_digests.json          sha256 of the parsed arrays of each shipped file (so the
                                                     equality can be re-checked where the reference is absent)
"""
import hashlib
import itertools
import json
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = os.environ.get("METAGYM_REFERENCE", "/root/reference")
ASSETS = os.path.join(REF, "metagym", "metalocomotion", "envs", "assets")

from metagym_amd.metalocomotion import variants  # noqa: E402
from metagym_amd.metalocomotion.mjcf import load_mjcf  # noqa: E402

DIGEST_KEYS = ("body_parent", "body_pos", "body_rot", "body_mass", "body_com", "body_inertia", "joint_body",
               "joint_anchor", "joint_axis", "joint_lo", "joint_hi", "joint_armature", "joint_damping",
               "joint_stiffness", "sph_body", "sph_pos", "sph_radius", "geom_body", "geom_p0", "geom_p1",
               "geom_radius", "pair_a", "pair_b", "geom_friction", "foot_body")


def model_digest(m):
    """sha256 over the numeric arrays of a parsed Model (fixed key order, C-contiguous little-endian bytes)
    and its body / joint names."""
    h = hashlib.sha256()
    d = m.to_dict()
    for k in DIGEST_KEYS:
        a = np.ascontiguousarray(d[k])
        h.update(k.encode() + str(a.dtype).encode() + str(a.shape).encode() + a.tobytes())
    h.update(",".join(m.body_names).encode() + b"|" + ",".join(m.joint_names).encode())
    return h.hexdigest()


def _nums(s):
    return [float(x) for x in s.split()]


def _candidates(pairs, width=6):
    """All doubles p near written / base with base * p == written for every (base, written) pair (base != 0)."""
    base, written = next((b, w) for b, w in pairs if b != 0.0)
    p0 = written / base
    cands = [p0]
    lo = hi = p0
    for _ in range(width):
        lo, hi = np.nextafter(lo, -np.inf), np.nextafter(hi, np.inf)
        cands += [float(lo), float(hi)]
    return [p for p in cands if all(b * p == w for b, w in pairs)]


def humanoid_pattern(path):
    base = variants.humanoid_spec()["root"]
    b_pelvis = variants._child_bodies(variants._child_bodies(base)[0])[0]
    root = ET.parse(path).getroot()
    torso = root.find("worldbody").find("body")
    pelvis = torso.find("body").find("body")
    pairs = [[], [], []]
    w = _nums(pelvis.find("geom").get("fromto"))
    b = variants._first_capsule(b_pelvis)["fromto"]
    pairs[0] += [(b[1], w[1]), (b[4], w[4])]
    for thigh, b_thigh in zip(pelvis.findall("body"), variants._child_bodies(b_pelvis)):
        shin, b_shin = thigh.find("body"), variants._child_bodies(b_thigh)[0]
        foot, b_foot = shin.find("body"), variants._child_bodies(b_shin)[0]
        pairs[0].append((b_thigh["pos"][1], _nums(thigh.get("pos"))[1]))
        pairs[1] += list(zip(variants._first_capsule(b_thigh)["fromto"], _nums(thigh.find("geom").get("fromto"))))
        pairs[1] += list(zip(b_shin["pos"], _nums(shin.get("pos"))))
        pairs[2] += list(zip(variants._first_capsule(b_shin)["fromto"], _nums(shin.find("geom").get("fromto"))))
        pairs[2] += list(zip(b_foot["pos"], _nums(foot.get("pos"))))
    cands = [_candidates(p) for p in pairs]
    z = _nums(torso.get("pos"))[2]
    for p in itertools.product(*cands):
        if base["pos"][2] + (0.403 * (p[1] - 1.0) + 0.45 * (p[2] - 1.0) - 0.20) == z:
            return list(p)
    raise RuntimeError("no pattern reproduces %s (candidates %s)" % (path, [len(c) for c in cands]))


def ant_pattern(path):
    base = variants.ant_spec()["root"]
    root = ET.parse(path).getroot()
    torso = root.find("worldbody").find("body")
    out = []
    for leg, b_leg in zip(torso.findall("body"), variants._child_bodies(base)):
        aux, b_aux = leg.find("body"), variants._child_bodies(b_leg)[0]
        foot, b_foot = aux.find("body"), variants._child_bodies(b_aux)[0]
        for seg, b_seg, nxt in ((leg, b_leg, aux), (aux, b_aux, foot), (foot, b_foot, None)):
            bf = variants._first_capsule(b_seg)["fromto"]
            pairs = list(zip(bf, _nums(seg.find("geom").get("fromto"))))
            if nxt is not None:
                pairs += list(zip(bf[3:], _nums(nxt.get("pos"))))
            c = _candidates(pairs)
            if not c:
                raise RuntimeError("no factor reproduces a segment of %s" % path)
            out.append(c[0])
    return out


def main():
    if not os.path.isdir(ASSETS):
        raise SystemExit("reference assets not found at %s — this script runs in the build container only" % ASSETS)
    out, digests = {}, {}
    for robot, sub, finder in (("humanoid", "humanoids", humanoid_pattern), ("ant", "ants", ant_pattern)):
        feet = variants.FEET[robot]
        digests["%s.xml" % robot] = model_digest(load_mjcf(os.path.join(ASSETS, sub, "%s.xml" % robot), foot_names=feet, preset="mujoco"))
        for tag, count in (("tra", 256), ("tst", 64), ("ood", 64)):
            pats = []
            for i in range(count):
                name = "%s_var_%s_%03d.xml" % (robot, tag, i)
                path = os.path.join(ASSETS, sub, name)
                p = finder(path)
                ref = load_mjcf(path, foot_names=feet, preset="mujoco")
                mine = load_mjcf(variants.mjcf_text(robot, p), foot_names=feet, preset="mujoco")
                rd, md = ref.to_dict(), mine.to_dict()
                for k in rd:
                    if not np.array_equal(rd[k], md[k]):
                        raise RuntimeError("%s: regenerated model differs in %s" % (name, k))
                pats.append(p)
                digests[name] = model_digest(ref)
            out["%s_%s" % (robot, tag)] = np.asarray(pats, np.float64)
            print(robot, tag, count, "variants reproduced exactly; pattern range",
                  out["%s_%s" % (robot, tag)].min(0).round(3)[:3], out["%s_%s" % (robot, tag)].max(0).round(3)[:3])
    np.savez_compressed(os.path.join(ROOT, "metagym_amd", "metalocomotion", "variant_patterns.npz"), **out)
    with open(os.path.join(ROOT, "tests", "golden", "walker_variant_digests.json"), "w") as f:
        json.dump({"numpy_version": np.__version__, "digests": digests}, f, indent=0, sort_keys=True)
    print("wrote variant_patterns.npz and walker_variant_digests.json (%d models)" % len(digests))


if __name__ == "__main__":
    main()
