#!/bin/bash
# Round 5, GPU call G: full GPU test suite on the current build; 256x256 A/B of the deferred frame store (3 repetitions each)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r05g; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log
M="python scripts/bench_maze.py --skip2d --no-u8 --res 256 --steps 30 --warmup 5"
for i in 1 2 3; do
  for v in default maze_pl6; do
    echo "== $v rep $i"
    if [ $v = default ]; then timeout 200 $M; else METAGYM_HIP_LIB=metagym_amd/lib/variants/$v.so timeout 200 $M; fi
  done
done > $OUT/maze3d_256_ab.txt 2>&1
for r in 64 32; do echo "== shipped $r"; timeout 120 python scripts/probe_maze3d_64.py $r; done > $OUT/maze3d_small.txt 2>&1
tail -3 $OUT/pytest_gpu.log; grep -v amdgpu.ids $OUT/maze3d_256_ab.txt | grep -o '^==.*\|"workload": "[a-z-]*3D\|avg_launch_ms": [0-9.]*' | paste - - - - - ; grep -v amdgpu.ids $OUT/maze3d_small.txt
