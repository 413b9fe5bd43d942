"""The kinematics-as-scans algorithm of the walker wave kernel (metagym_amd/csrc/walker.hip: wave_kinematics and the scan
tables its prologue builds), restated in numpy and checked against the plain serial recursion on random kinematic trees —
no GPU needed. What is pinned here is the ALGORITHM and its table / round-count logic (the 2^r-th-ancestor tables over
"hops", the joint tree, ceil(log2) round counts as mg_walker_step computes them); the device code itself is checked against
the oracle on the GPU by tests/test_walker_gpu.py and tests/test_walker_generic_gpu.py."""
import zlib

import numpy as np
import pytest


def _random_tree(rs, nb, max_joints_per_body=3, nj_cap=24):
    parent = [-1] + [int(rs.randint(0, b)) for b in range(1, nb)]            # parents come first
    counts = [0] + [int(rs.randint(0, max_joints_per_body + 1)) for _ in range(1, nb)]
    while sum(counts) > nj_cap:
        counts[int(np.argmax(counts))] -= 1
    joint_body = [b for b in range(nb) for _ in range(counts[b])]            # non-decreasing
    return parent, joint_body


def _tables(parent, joint_body):
    """The prologue's tables, written the way the kernel writes them (lane = hop / joint / body)."""
    nb, nj = len(parent), len(joint_body)
    jstart, jcount = [0] * nb, [0] * nb
    for j, b in enumerate(joint_body):
        if j == 0 or joint_body[j - 1] != b:
            jstart[b] = j
        jcount[b] += 1
    H = nb + nj
    hp, hbody = [-1] * H, [-1] * H
    for b in range(nb):
        pb = parent[b]
        if pb >= 0:
            hp[b] = nb + jstart[pb] + jcount[pb] - 1 if jcount[pb] > 0 else pb
        hbody[b] = b if jcount[b] == 0 else -1
    for j, b in enumerate(joint_body):
        hp[nb + j] = nb + j - 1 if j > jstart[b] else b
        hbody[nb + j] = b if j == jstart[b] + jcount[b] - 1 else -1

    def joint_at_or_above(h):
        while 0 <= h < nb:
            pb = parent[h]
            h = -1 if pb < 0 else (nb + jstart[pb] + jcount[pb] - 1 if jcount[pb] > 0 else pb)
        return -1 if h < 0 else h - nb
    jprev = [joint_at_or_above(hp[nb + j]) for j in range(nj)]
    bjoint = [jstart[b] + jcount[b] - 1 if jcount[b] > 0 else joint_at_or_above(hp[b]) for b in range(nb)]
    # round counts as the host computes them (mg_walker_step)
    hops, joints = [0] * nb, [0] * nb
    for b in range(nb):
        pb = parent[b]
        hops[b] = (0 if pb < 0 else hops[pb]) + 1 + jcount[b]
        joints[b] = (0 if pb < 0 else joints[pb]) + jcount[b]
    rh = 0
    while (1 << rh) < max(hops):
        rh += 1
    jr = 0
    while (1 << jr) < max(joints):
        jr += 1
    if nj > 0 and jr < 1:
        jr = 1
    hanc = [hp]
    for r in range(1, rh):
        hanc.append([hanc[r - 1][a] if a >= 0 else -1 for a in hanc[r - 1]])
    janc = [jprev]
    for r in range(1, jr):
        janc.append([janc[r - 1][a] if a >= 0 else -1 for a in janc[r - 1]])
    return dict(nb=nb, nj=nj, jstart=jstart, jcount=jcount, hbody=hbody, hanc=hanc[:max(rh, 0)], janc=janc, bjoint=bjoint,
                rh=rh, jr=jr)


def _rodrigues(k, t):
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(t) * K + (1 - np.cos(t)) * K @ K


def _serial(parent, joint_body, body_rot, body_pos, anchor, axis, q, qd, base_R, base_p, base_v, base_w):
    """Body by body, joint by joint (the lane kernel / oracle order)."""
    nb = len(parent)
    R, o = [None] * nb, [None] * nb
    fw, fal, fxr, far, fvr = ([None] * nb for _ in range(5))
    p, a = {}, {}
    j = 0
    for b in range(nb):
        pb = parent[b]
        if pb < 0:
            Rc, oc, w, al, xr, ar, vr = base_R.copy(), base_p.copy(), base_w.copy(), np.zeros(3), base_p.copy(), np.zeros(3), base_v.copy()
        else:
            Rc, oc = R[pb] @ body_rot[b], o[pb] + R[pb] @ body_pos[b]
            w, al, xr, ar, vr = fw[pb], fal[pb], fxr[pb], far[pb], fvr[pb]
        while j < len(joint_body) and joint_body[j] == b:
            pj, aj = oc + Rc @ anchor[j], Rc @ axis[j]
            p[j], a[j] = pj, aj
            Rn = Rc @ _rodrigues(axis[j], q[j])
            oc = pj - Rn @ anchor[j]
            Rc = Rn
            r = pj - xr
            ar = ar + np.cross(al, r) + np.cross(w, np.cross(w, r))
            vr = vr + np.cross(w, r)
            xr = pj
            wj = qd[j] * aj
            al = al + np.cross(w, wj)
            w = w + wj
            j += 1
        R[b], o[b], fw[b], fal[b], fxr[b], far[b], fvr[b] = Rc, oc, w, al, xr, ar, vr
    return R, o, p, a, fw, fal, fxr, far, fvr


def _scans(T, body_rot, body_pos, anchor, axis, q, qd, base_R, base_p, base_v, base_w):
    """wave_kinematics: lane = hop; rh rounds over the hop tables, then three scans over the joint tree (jr rounds)."""
    nb, nj, H = T["nb"], T["nj"], T["nb"] + T["nj"]
    Rh, th = [None] * H, [None] * H
    for h in range(H):
        if h >= nb:
            j = h - nb
            Rh[h] = _rodrigues(axis[j], q[j])
            th[h] = anchor[j] - Rh[h] @ anchor[j]
        elif h == 0:
            Rh[h], th[h] = base_R.copy(), base_p.copy()
        else:
            Rh[h], th[h] = body_rot[h].copy(), body_pos[h].copy()
    for r in range(T["rh"]):
        Rn, tn = list(Rh), list(th)
        for h in range(H):                                  # all lanes read the previous round's values, then write
            a = T["hanc"][r][h]
            if a >= 0:
                Rn[h], tn[h] = Rh[a] @ Rh[h], Rh[a] @ th[h] + th[a]
        Rh, th = Rn, tn
    R, o, p, a = [None] * nb, [None] * nb, {}, {}
    for h in range(H):
        if h >= nb:
            p[h - nb], a[h - nb] = th[h] + Rh[h] @ anchor[h - nb], Rh[h] @ axis[h - nb]
        if T["hbody"][h] >= 0:
            R[T["hbody"][h]], o[T["hbody"][h]] = Rh[h], th[h]

    def scan(vals):
        v = [x.copy() for x in vals]
        for r in range(T["jr"]):
            prev = [x.copy() for x in v]
            for j in range(nj):
                an = T["janc"][r][j]
                if an >= 0:
                    v[j] = v[j] + prev[an]
        return v
    jprev = T["janc"][0] if nj else []
    wj = [qd[j] * a[j] for j in range(nj)]
    Sw = scan(wj)
    wbef = [base_w + (Sw[jprev[j]] if jprev[j] >= 0 else 0) for j in range(nj)]
    rj = [p[j] - (p[jprev[j]] if jprev[j] >= 0 else base_p) for j in range(nj)]
    Sal = scan([np.cross(wbef[j], wj[j]) for j in range(nj)])
    Svr = scan([np.cross(wbef[j], rj[j]) for j in range(nj)])
    albef = [Sal[jprev[j]] if jprev[j] >= 0 else np.zeros(3) for j in range(nj)]
    Sar = scan([np.cross(albef[j], rj[j]) + np.cross(wbef[j], np.cross(wbef[j], rj[j])) for j in range(nj)])
    fw, fal, fxr, far, fvr = ([None] * nb for _ in range(5))
    for b in range(nb):
        k = T["bjoint"][b]
        fw[b] = base_w + (Sw[k] if k >= 0 else 0)
        fal[b] = Sal[k] if k >= 0 else np.zeros(3)
        fxr[b] = p[k] if k >= 0 else base_p
        far[b] = Sar[k] if k >= 0 else np.zeros(3)
        fvr[b] = base_v + (Svr[k] if k >= 0 else 0)
    return R, o, p, a, fw, fal, fxr, far, fvr


def _random_rotation(rs):
    k = rs.normal(size=3)
    return _rodrigues(k / np.linalg.norm(k), rs.uniform(-np.pi, np.pi))


CASES = [("single body", [-1], []), ("legs one joint deep", [-1, 0, 0, 0, 0], [1, 2, 3, 4]),
         ("jointless bodies between jointed ones", [-1, 0, 1, 2, 3], [2, 4, 4]),
         ("snake", [-1] + list(range(15)), list(range(1, 16))),
         ("humanoid", [-1, 0, 1, 2, 3, 4, 2, 6, 7, 0, 9, 0, 11], [1, 1, 2, 3, 3, 3, 4, 6, 6, 6, 7, 9, 9, 10, 11, 11, 12]),
         ("ant", [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11], [2, 3, 5, 6, 8, 9, 11, 12])]


@pytest.mark.parametrize("case", CASES + [("random %d" % s, None, None) for s in range(40)], ids=lambda c: c[0])
def test_scans_equal_the_serial_recursion(case):
    name, parent, joint_body = case
    rs = np.random.RandomState(zlib.crc32(name.encode()))
    if parent is None:
        parent, joint_body = _random_tree(rs, int(rs.randint(1, 17)))
    nb, nj = len(parent), len(joint_body)
    T = _tables(parent, joint_body)
    # round counts are the smallest that cover the longest chain (and the joint table always has its first round)
    assert T["rh"] <= 6 and (nj == 0 or T["jr"] >= 1)
    body_rot = [_random_rotation(rs) for _ in range(nb)]
    body_pos = [rs.uniform(-0.5, 0.5, 3) for _ in range(nb)]
    anchor = [rs.uniform(-0.3, 0.3, 3) for _ in range(nj)]
    axis = [(lambda k: k / np.linalg.norm(k))(rs.normal(size=3)) for _ in range(nj)]
    q, qd = rs.uniform(-1.5, 1.5, nj), rs.uniform(-8, 8, nj)
    base = (_random_rotation(rs), rs.uniform(-1, 1, 3), rs.uniform(-2, 2, 3), rs.uniform(-3, 3, 3))
    ser = _serial(parent, joint_body, body_rot, body_pos, anchor, axis, q, qd, *base)
    sc = _scans(T, body_rot, body_pos, anchor, axis, q, qd, *base)
    for s_, c_, what in zip(ser, sc, ("R", "o", "p", "a", "w", "alpha", "x_ref", "a_ref", "v_ref")):
        keys = range(nb) if isinstance(s_, list) else sorted(s_)
        for k in keys:
            assert np.allclose(s_[k], c_[k], rtol=0, atol=1e-10), (name, what, k, np.abs(s_[k] - c_[k]).max())
