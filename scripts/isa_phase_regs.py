"""Per-phase VGPR pressure of the walker wave kernel, from its gfx950 assembly (no GPU needed).

    python scripts/isa_phase_regs.py [kernel-name substring, default the damped humanoid instantiation]

Compiles metagym_amd/csrc/walker.hip with -DMG_WALKER_PHASE_MARKS (PHASE(i) becomes an assembly comment, nothing else changes) to
/tmp/isa/walker_marks.s, cuts the chosen kernel at the markers and reports, per phase of the sub-step:
  instructions (VALU / LDS / VMEM / SALU), distinct VGPRs referenced, the highest VGPR index referenced, and the peak number of
  simultaneously LIVE VGPRs from a backward liveness pass over the instructions in program order.
The liveness pass ignores control flow (the sub-step is almost entirely straight-line, unrolled code; a value carried around a loop's
back edge is counted from its first definition to its last use in program order), treats v_writelane as read-modify-write, and knows
which mnemonics define their first operand. It is an estimate, good to a few registers — enough to say WHICH phase pins the kernel's
allocation (the allocator's count is the peak over all phases)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {"begin": "actuators / prologue", 11: "kinematics level loop", 0: "kinematics", 1: "body inertia + subtree sums", 2: "S_d, F_d, h and M entries",
         3: "Cholesky", 4: "free motion", 5: "detection + selection", 6: "contact rows", 7: "whitening (+ A = Jh Jh^T rows)", 8: "PGS",
         9: "back-solve + integrate"}


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return [int(m.group(1))]
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


NO_DEF = ("ds_write", "ds_store", "global_store", "buffer_store", "flat_store", "scratch_store", "v_cmp", "v_readlane", "v_readfirstlane",
          "s_", "buffer_wbl2", "buffer_inv", "global_atomic", "ds_add", "ds_or", "v_nop", "ds_bpermute_fake")


def parse(line):
    """-> (mnemonic, defs, uses) of one instruction line, or None"""
    line = line.split(";")[0].strip()
    if not line or line.endswith(":") or line.startswith("."):
        return None
    parts = line.split(None, 1)
    mn = parts[0]
    ops = []
    if len(parts) > 1:
        for tok in re.split(r",\s*|\s+", parts[1]):
            if tok.startswith("v") and regs(tok):
                ops.append(regs(tok))
    defs, uses = [], []
    if not ops:
        return mn, defs, uses
    has_def = not mn.startswith(NO_DEF) or mn.startswith("v_cmpx") is False and False
    if mn.startswith(NO_DEF):
        has_def = False
    if mn.startswith("global_atomic") or mn.startswith("ds_add") or mn.startswith("ds_or"):
        has_def = "rtn" in mn or "_ret" in mn
    if has_def:
        defs = ops[0]
        for o in ops[1:]:
            uses += o
        if mn.startswith("v_writelane") or mn.startswith("v_mac") or mn.startswith("v_fmac") or "dpp" in line and "bound_ctrl" not in line:
            uses += ops[0]          # read-modify-write / old value kept
    else:
        for o in ops:
            uses += o
    return mn, defs, uses


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else "ILi23ENS_5ShapeILi13ELi17ELi29ELi17ELi1ELi1E"
    os.makedirs("/tmp/isa", exist_ok=True)
    out = "/tmp/isa/walker_marks.s"
    src = os.path.join(ROOT, "metagym_amd", "csrc", "walker.hip")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                               "--cuda-device-only", "-S", "-DMG_WALKER_PHASE_MARKS", "-I", os.path.join(ROOT, "include"), src, "-o", out],
                              stderr=subprocess.DEVNULL)
    text = open(out).read()
    # the kernel's body: from its label to s_endpgm
    m = re.search(r"^(_ZN[^\n:]*%s[^\n:]*):[^\n]*\n" % re.escape(want), text, re.M)
    if not m:
        raise SystemExit("no kernel matching %r" % want)
    body = text[m.end():]
    body = body[:body.index("s_endpgm")]
    vg = re.search(r"\.vgpr_count:\s+(\d+)", text[text.index(m.group(1), text.index(".amdhsa_kernel") if ".amdhsa_kernel" in text else 0):] if False else text)
    lines = body.split("\n")
    insts, phase_of, marks = [], [], []
    cur = "(before the first marker)"
    for ln in lines:
        mm = re.search(r"; MGPHASE (\w+)", ln)
        if mm:
            marks.append((len(insts), mm.group(1)))
            continue
        p = parse(ln)
        if p:
            insts.append(p)
    # a marker CLOSES a phase (PHASE(i) is placed at the END of phase i); "begin" opens the sub-step
    seg_name = [None] * len(insts)
    start = 0
    last = "(kernel prologue / epilogue)"
    bounds = []
    for pos, tag in marks:
        name = NAMES.get(int(tag) if tag.isdigit() else tag, tag)
        if tag == "begin":
            bounds.append((start, pos, "(outside the sub-step)"))
        else:
            bounds.append((start, pos, name))
        start = pos
    bounds.append((start, len(insts), "(after the last marker: calc_state, stores)"))
    # backward liveness in program order
    live = set()
    live_at = [0] * len(insts)
    for i in range(len(insts) - 1, -1, -1):
        mn, defs, uses = insts[i]
        live -= set(defs)
        live |= set(uses)
        live_at[i] = len(live)
    print("%-44s %6s %6s %5s %5s %6s | %9s %8s %9s" % ("phase (in program order)", "VALU", "LDS", "VMEM", "SALU", "other", "VGPR refd", "max idx", "peak live"))
    agg = {}
    for a, b, name in bounds:
        if b <= a:
            continue
        seg = insts[a:b]
        cnt = {"VALU": 0, "LDS": 0, "VMEM": 0, "SALU": 0, "other": 0}
        refd = set()
        for mn, defs, uses in seg:
            k = "VALU" if mn.startswith("v_") else "LDS" if mn.startswith("ds_") else "VMEM" if mn.startswith(("global_", "buffer_", "flat_", "scratch_")) else \
                "SALU" if mn.startswith("s_") else "other"
            cnt[k] += 1
            refd |= set(defs) | set(uses)
        peak = max(live_at[a:b])
        row = agg.setdefault(name, dict(cnt=dict.fromkeys(cnt, 0), refd=set(), peak=0, segs=0))
        for k in cnt:
            row["cnt"][k] += cnt[k]
        row["refd"] |= refd
        row["peak"] = max(row["peak"], peak)
        row["segs"] += 1
    for name, row in agg.items():
        c = row["cnt"]
        print("%-44s %6d %6d %5d %5d %6d | %9d %8s %9d%s" % (name[:44], c["VALU"], c["LDS"], c["VMEM"], c["SALU"], c["other"], len(row["refd"]),
                                                            max(row["refd"]) if row["refd"] else "-", row["peak"], "  (x%d code copies)" % row["segs"] if row["segs"] > 1 else ""))
    print("total instructions %d; overall peak live %d; highest VGPR index referenced %d" % (len(insts), max(live_at), max(max(d + u, default=0) for _, d, u in insts)))


if __name__ == "__main__":
    main()
