#!/bin/bash
# Round 5, GPU call C: maze3d — the SMALL instantiation (crossbar broadcast + deferred frame store), the pointer-select fix, the
# deferred store in the multi-wave kernels (variants); bench K = 20 with the warm-up next to the timed region
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r05c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_maze_gpu.py tests/test_mixed_gpu.py tests/test_graph_capture_gpu.py tests/test_multi_device_gpu.py -m gpu -x -q > $OUT/pytest_maze.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_maze.log
METAGYM_HIP_LIB=metagym_amd/lib/variants/maze_pl5.so timeout 900 python -m pytest tests/test_maze_gpu.py -m gpu -x -q > $OUT/pytest_maze_pl5.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_maze_pl5.log
for r in 64 32; do
  echo "== shipped (SMALL: crossbar broadcast + deferred store) $r"; timeout 120 python scripts/probe_maze3d_64.py $r
  echo "== MG_MAZE3D_NO_SMALL=1 (general kernel) $r"; MG_MAZE3D_NO_SMALL=1 timeout 120 python scripts/probe_maze3d_64.py $r
  echo "== shipped, MG_MAZE3D_WAVES=1,64 (64-column slabs) $r"; MG_MAZE3D_WAVES=1,64 timeout 120 python scripts/probe_maze3d_64.py $r
done > $OUT/maze3d_small_frames.txt 2>&1
M="python scripts/bench_maze.py --skip2d --no-u8 --res 256 --steps 30 --warmup 5"
for v in default maze_pl6 maze_pl5 maze_w5; do
  echo "== $v"
  if [ $v = default ]; then timeout 200 $M; else METAGYM_HIP_LIB=metagym_amd/lib/variants/$v.so timeout 200 $M; fi
done > $OUT/maze3d_256_variants.txt 2>&1
for v in default maze_pl5; do
  echo "== $v 128"
  if [ $v = default ]; then timeout 100 python scripts/probe_maze3d_64.py 128; else METAGYM_HIP_LIB=metagym_amd/lib/variants/$v.so timeout 100 python scripts/probe_maze3d_64.py 128; fi
done > $OUT/maze3d_128.txt 2>&1
for i in 1 2 3; do timeout 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-pmc; done > $OUT/bench_20steps_x3.jsonl 2> $OUT/bench_20steps_x3.err
tail -3 $OUT/pytest_maze.log; tail -3 $OUT/pytest_maze_pl5.log; grep -v amdgpu.ids $OUT/maze3d_small_frames.txt; grep -v amdgpu.ids $OUT/maze3d_256_variants.txt | cut -c1-250; grep -v amdgpu.ids $OUT/maze3d_128.txt; grep -o '"ms_per_step": [0-9.]*' $OUT/bench_20steps_x3.jsonl
