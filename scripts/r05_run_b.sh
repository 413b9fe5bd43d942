#!/bin/bash
# Round 5, GPU call B: the small-frame maze renderer (parity + A/B timing), host cost of Quadrotor.step, walker third-wave experiments
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r05b; mkdir -p $OUT
timeout 900 python -m pytest tests/test_maze_gpu.py tests/test_mixed_gpu.py tests/test_graph_capture_gpu.py -m gpu -x -q > $OUT/pytest_maze.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_maze.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
for r in 64 32; do
  echo "== TRANS $r"; timeout 120 python scripts/probe_maze3d_64.py $r
  echo "== general path (MG_MAZE3D_NO_TRANS=1) $r"; MG_MAZE3D_NO_TRANS=1 timeout 120 python scripts/probe_maze3d_64.py $r
done > $OUT/maze3d_small_frames.txt 2>&1
timeout 120 python scripts/quad_host_cost.py > $OUT/quad_host_cost.txt 2>&1
for i in 1 2 3; do timeout 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-pmc; done > $OUT/bench_20steps_x3.jsonl 2> $OUT/bench_20steps_x3.err
for v in default walker_e1 walker_e1b walker_e2 walker_e3; do
  echo "== $v"
  if [ $v = default ]; then timeout 200 python scripts/bench_walker.py humanoid; else METAGYM_HIP_LIB=metagym_amd/lib/variants/$v.so timeout 200 python scripts/bench_walker.py humanoid; fi
done > $OUT/walker_third_wave.txt 2>&1
tail -3 $OUT/pytest_maze.log; cat $OUT/smoke.log | tail -2; cat $OUT/maze3d_small_frames.txt $OUT/quad_host_cost.txt; grep -o '"ms_per_step": [0-9.]*' $OUT/bench_20steps_x3.jsonl; grep -v amdgpu.ids $OUT/walker_third_wave.txt | cut -c1-300
