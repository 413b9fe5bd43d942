#!/bin/bash
# Differential fuzzers HIP vs oracle with this round's seeds (GPU box, through gpurun):  scripts/fuzz_round.sh r03 41
cd "${GRAFT_REPO_ROOT:-.}"
R=${1:-r03}; S=${2:-41}; OUT=gpurun_out/$R; mkdir -p $OUT
for job in "quadrotor 200" "maze3d 160" "maze2d 60" "sampler 100" "a1 200"; do
  set -- $job
  echo "scripts/fuzz_$1.py --configs $2 --seed $S" > $OUT/fuzz_$1.txt
  timeout 400 python scripts/fuzz_$1.py --configs $2 --seed $S 2>&1 | tail -2 >> $OUT/fuzz_$1.txt
done
tail -n 3 $OUT/fuzz_*.txt
