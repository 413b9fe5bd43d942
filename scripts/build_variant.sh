#!/bin/bash
# A/B builds for timing on the GPU box:  scripts/build_variant.sh NAME [GITREV] [extra hipcc flags...]
# compiles the csrc of GITREV (default: the working tree) into metagym_amd/lib/variants/NAME.so; pick it at run
# time with METAGYM_HIP_LIB=metagym_amd/lib/variants/NAME.so (the .so is git-ignored but travels with gpurun).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
rev=${1:-WORK}; [ $# -gt 0 ] && shift
out=metagym_amd/lib/variants; mkdir -p $out
if [ "$rev" = WORK ]; then src=metagym_amd/csrc; else
  tmp=/tmp/variant_$name; rm -rf $tmp; mkdir -p $tmp/metagym_amd/csrc $tmp/include
  for f in $(git ls-tree --name-only $rev metagym_amd/csrc/); do git show $rev:$f > $tmp/$f; done
  git show $rev:include/metagym_hip.h > $tmp/include/metagym_hip.h
  src=$tmp/metagym_amd/csrc
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
  -Wno-unused-function "$@" $src/*.hip -o $out/$name.so
ls -la $out/$name.so
