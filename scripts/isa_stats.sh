#!/bin/bash
# Register / spill / LDS metadata of every kernel in one .hip file (gfx950 code object, no GPU needed):
#   scripts/isa_stats.sh metagym_amd/csrc/quadrotor.hip [extra hipcc flags]
# The assembly lands in /tmp/isa/<name>.s for reading.
set -e
src=$1; shift
name=$(basename "$src" .hip)
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math --cuda-device-only -S \
    "$@" "$src" -o /tmp/isa/$name.s
python3 - "$name" <<'PY'
import re, sys
txt = open("/tmp/isa/%s.s" % sys.argv[1]).read()
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", txt, re.S):
    pass
blocks = re.split(r"\n  - \.agpr_count", txt)
for b in blocks[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, b) or [None, "?"])[1]
    print("%-90s vgpr %-4s sgpr %-4s sgpr_spill %-4s vgpr_spill %-4s lds %-6s scratch %s" % (
        g("name")[:90], g("vgpr_count"), g("sgpr_count"), g("sgpr_spill_count"), g("vgpr_spill_count"),
        g("group_segment_fixed_size"), g("private_segment_fixed_size")))
PY
